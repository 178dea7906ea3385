// include/btlbf/detail.hpp -- helpers shared by the drop-in C++ shims over the C ABI (btlbf.h).
//
// The shims keep the reference's error convention: a failed call prints a message on std::cerr
// and terminates the process with exit(1) (reference: BloomFilter.hpp:124-129,144-148,391-394;
// vendor/IOUtil.h:14-22).  Define BTLBF_SHIM_THROW to get std::runtime_error instead.
#ifndef BTLBF_DETAIL_HPP
#define BTLBF_DETAIL_HPP
#include "../btlbf.h"

#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>

namespace btlbf_shim {

inline int&
default_device()
{
	static int dev = [] {
		const char* e = std::getenv("BTLBF_DEVICE");
		return e ? std::atoi(e) : 0;
	}();
	return dev;
}

inline void
check(int rc)
{
	if (rc == BTLBF_OK)
		return;
#ifdef BTLBF_SHIM_THROW
	throw std::runtime_error(btlbf_last_error());
#else
	std::cerr << btlbf_last_error() << std::endl;
	std::exit(EXIT_FAILURE);
#endif
}

inline bool
bit(const uint64_t* words, uint64_t p)
{
	return (words[p >> 6] >> (p & 63)) & 1u;
}

} // namespace btlbf_shim
#endif
