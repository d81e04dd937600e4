// crc32_fold_check.cpp -- TEST INFRASTRUCTURE: rsem_amd/csrc/host/crc32_fold.hpp (CRC-32 by carry-less multiplication) against zlib's crc32 on
// every length 0 .. 400 at three alignments, on 2 000 random (length, start, seed) triples, and the two rates.  Exit code 0 = all equal.
#include "../rsem_amd/csrc/host/crc32_fold.hpp"
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
int main(){ std::mt19937_64 r(1); std::vector<unsigned char> b(1<<20); for(auto&x:b)x=(unsigned char)r(); int bad=0;
 for(size_t n=0;n<=400;n++) for(size_t o=0;o<3;o++){ uint32_t a=crc32(0,b.data()+o,n), c=rsemh::crc32_fast(0,b.data()+o,n); if(a!=c){ if(bad<5) printf("n=%zu o=%zu zlib %08x mine %08x\n",n,o,a,c); bad++; } }
 for(int t=0;t<2000;t++){ size_t n=r()%70000, o=r()%1000; uint32_t s=(uint32_t)r(); uint32_t a=crc32(s,b.data()+o,n), c=rsemh::crc32_fast(s,b.data()+o,n); if(a!=c){ if(bad<5) printf("n=%zu seed %08x zlib %08x mine %08x\n",n,s,a,c); bad++; } }
 printf("bad %d usable %d\n",bad,(int)rsemh::crc32_fold_usable());
 auto t0=std::chrono::steady_clock::now(); uint32_t x=0; for(int k=0;k<2000;k++) x^=rsemh::crc32_fast(0,b.data(),65280); double s1=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
 t0=std::chrono::steady_clock::now(); for(int k=0;k<2000;k++) x^=crc32(0,b.data(),65280); double s2=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
 printf("fold %.0f MB/s, zlib %.0f MB/s (%u)\n",2000*65280/1e6/s1,2000*65280/1e6/s2,x); return bad!=0; }
