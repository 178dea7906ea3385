// csrc/host_internal.hpp -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include "../../include/btlbf.h"

// set the thread-local message behind btlbf_last_error() and return `code` (capi.cpp)
int btlbf_set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
