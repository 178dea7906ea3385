/* bindings/swig/BloomFilter.i -- SWIG interface of the MI355X-native drop-in, for Perl5 (or any SWIG target).
 *
 * Module name, class name and method set are those of the reference's binding
 * (/root/reference/swig/BloomFilter.i:1-59: KmerBloomFilter exposed as `BloomFilter`, plus insertSeq), so the
 * reference's scripts (swig/test.pl, swig/writeBloom_rolling.pl, swig/testBloom_rolling.pl) run unchanged:
 *     use BloomFilter;
 *     $f = BloomFilter::BloomFilter->new($bits, $h, $k);  $f->insert("ACGT...");  $f->contains("ACGT...");
 *     BloomFilter::insertSeq($f, $seq, $h, $k);           $f->storeFilter("x.bf");
 * What is wrapped are the header-only shims in include/btlbf/ over the C ABI (include/btlbf.h): every call
 * ends in a HIP kernel; the wrapper library links libbtlbf.so.  Added to the reference's surface: the whole-
 * sequence query the GPU wants instead of one call per k-mer (containsSeq, countHits).
 *
 * Build (swig is not part of the ROCm image this repository is developed in; see README.md in this directory):
 *     swig -Wall -c++ -perl5 -I../../include BloomFilter.i
 *     g++ -std=c++17 -fPIC -c BloomFilter_wrap.cxx -I../../include $(perl -MConfig -e 'print "-I$Config{archlib}/CORE"')
 *     g++ -shared BloomFilter_wrap.o -L../../btl_bloomfilter_amd -lbtlbf -Wl,-rpath,'$ORIGIN/../../btl_bloomfilter_amd' -o BloomFilter.so
 */
%module BloomFilter
%include "std_string.i"
%include "stdint.i"
%include "std_vector.i"
namespace std {
   %template(SizetVector) vector<size_t>;
   %template(Uint64Vector) vector<uint64_t>;
   %template(BoolVector) vector<bool>;
}

%{
#include "btlbf/KmerBloomFilter.hpp"
#include "btlbf/ntHashIterator.hpp"
#include "btlbf/BloomFilterUtil.h"
%}

%rename(BloomFilter) KmerBloomFilter;

using namespace std;

/* reference: swig/BloomFilter.i:21-40 (KmerBloomFilter.hpp:17-75 over BloomFilter.hpp:46-381) */
class KmerBloomFilter {
public:
        KmerBloomFilter();
        ~KmerBloomFilter();
        KmerBloomFilter(uint64_t filterSize, unsigned hashNum, unsigned kmerSize);
        KmerBloomFilter(const string &filterFilePath);

        void insert(vector<uint64_t> const &precomputed);
        void insert(const char* kmer);

        bool contains(vector<uint64_t> const &values);
        bool contains(const char* kmer);

        void storeFilter(string const &filterFilePath);
        uint64_t getPop();
        unsigned getHashNum();
        unsigned getKmerSize();
        uint64_t getFilterSize();
};

/* batch members of the drop-in (no counterpart in the reference): one kernel launch per sequence */
%extend KmerBloomFilter {
        /* every k-mer of seq, as BloomFilterUtil.h:9-17 does with an ntHashIterator loop */
        void insertSeq(const string& seq) { $self->BloomFilter::insertSeq(seq); }
        /* number of k-mers of seq found in the filter (what swig/testBloom_rolling.cpp:21-31 counts in a loop) */
        uint64_t countHits(const string& seq)
        {
                vector<bool> res, valid;
                $self->containsSeq(seq, res, valid);
                uint64_t n = 0;
                for (size_t i = 0; i < res.size(); ++i)
                        n += res[i];
                return n;
        }
}

/* reference: swig/BloomFilter.i:59 */
void insertSeq(KmerBloomFilter &bloom, const string& seq, unsigned numHashes, unsigned k);
