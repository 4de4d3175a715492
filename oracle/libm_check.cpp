// libm_check.cpp -- TEST INFRASTRUCTURE: compares strelka_amd/csrc/libm_flt32.h (the restatement of glibc's powf / logf that
// the kernels use for the reference's std::pow(float,float) / std::log(float) calls) with the host libm, bit for bit, on N
// pseudo-random arguments of the domain the path uses.  Built and run by tests/test_libm_restatement.py.
#include "../strelka_amd/csrc/libm_flt32.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
int main(int argc,char**argv){
  long n=atol(argv[1]); unsigned long long st=88172645463325252ull; long badp=0,badl=0,fall=0;
  for(long it=0;it<n;++it){
    st^=st<<13; st^=st>>7; st^=st<<17;
    int q=3+(st%68);
    float e=(float)std::pow(10.0,-0.1*q);
    if(it&1){ uint32_t u=sk_libm::as_u32(e); u+=(st>>20)%2048; e=sk_libm::as_f32(u);}
    float v=(float)(((st>>32)%1000000)/1000000.0); if(v<=0)v=0.25f;
    float a=powf(e,v), b=0; if(!sk_libm::powf_glibc(e,v,b)) {fall++; b=a;}
    if(sk_libm::as_u32(a)!=sk_libm::as_u32(b)) badp++;
    float x=sk_libm::as_f32(0x33000000u+(uint32_t)((st>>8)%0x0c800000u));
    float la=logf(x), lb=0; if(!sk_libm::logf_glibc(x,lb)) {fall++; lb=la;}
    if(sk_libm::as_u32(la)!=sk_libm::as_u32(lb)) badl++;
  }
  printf("n=%ld powf mismatches %ld logf mismatches %ld fallbacks %ld\n",n,badp,badl,fall);
  return badp||badl;
}
