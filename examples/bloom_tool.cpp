// examples/bloom_tool.cpp -- the reference's two command-line habits (build a filter from a FASTA/FASTQ
// file and store it; load a filter and query a file) on top of the drop-in headers.
//
//   g++ -std=c++17 -O2 -Iinclude examples/bloom_tool.cpp -Lbtl_bloomfilter_amd -lbtlbf \
//       -Wl,-rpath,$PWD/btl_bloomfilter_amd -Wl,-rpath,/opt/rocm/lib -o bloom_tool
//   ./bloom_tool build reads.fq.gz 549755813888 4 31 reads.bf      # 2^39 bits, h = 4, k = 31
//   ./bloom_tool query reads.bf contigs.fa
//
// Compare swig/writeBloom_rolling.cpp in the reference (contigsToBloom + storeFilter): the per-contig
// insertSeq loop is replaced by one insertFile call that parses, copies and hashes double-buffered.
#include "btlbf/BloomFilter.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static int usage()
{
	std::fprintf(stderr, "usage: bloom_tool build <fasta|fastq[.gz]> <bits> <hashes> <k> <out.bf>\n"
	                     "       bloom_tool query <filter.bf> <fasta|fastq[.gz]>\n");
	return 2;
}

int main(int argc, char** argv)
{
	if (argc < 2)
		return usage();
	if (!std::strcmp(argv[1], "build") && argc == 7) {
		BloomFilter bloom(std::strtoull(argv[3], nullptr, 10), (unsigned)std::atoi(argv[4]), (unsigned)std::atoi(argv[5]));
		const btlbf_fastx_stats st = bloom.insertFile(argv[2]);
		bloom.storeFilter(argv[6]);
		std::printf("records %llu  bases %llu  batches %llu  parse %.3f s  total %.3f s  popcount %llu  FPR %.3g\n",
		            (unsigned long long)st.n_records, (unsigned long long)st.n_bases, (unsigned long long)st.n_batches,
		            st.seconds_parse, st.seconds_total, (unsigned long long)bloom.getPop(), bloom.getFPR());
		return 0;
	}
	if (!std::strcmp(argv[1], "query") && argc == 4) {
		BloomFilter bloom{std::string(argv[2])};
		const btlbf_fastx_stats st = bloom.containsFile(argv[3]);
		std::printf("k-mers %llu  found %llu (%.2f %%)  total %.3f s\n", (unsigned long long)st.n_windows,
		            (unsigned long long)st.n_hits, st.n_windows ? 100.0 * (double)st.n_hits / (double)st.n_windows : 0.0,
		            st.seconds_total);
		return 0;
	}
	return usage();
}
