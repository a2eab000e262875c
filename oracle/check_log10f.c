/* check_log10f.c -- TEST INFRASTRUCTURE.  saveQval (Genrich.c:221, 226) calls the host libm's
 * log10f; the device restates glibc 2.35's algorithm (fdlibm log10f around the table-driven
 * logf).  This program checks that restatement against the libm actually installed, over every
 * positive normal float (stride 1, ~70 s) or a strided sample:  ./check_log10f [stride]
 * Exit status 0 iff there is no mismatch. */
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline uint32_t asuint(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float asfloat(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static const struct {double invc, logc;} T[16] = {
  { 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 },
  { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 },
  { 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2 },
  { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 },
  { 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 },
  { 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3 },
  { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 },
  { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 },
  { 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 },
  { 0x1p+0, 0x0p+0 },
  { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 },
  { 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4 },
  { 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 },
  { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3 },
  { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 },
  { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 },
};
static const double Ln2 = 0x1.62e42fefa39efp-1;
static const double A[3] = { -0x1.00ea348b88334p-2, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2 };
#define OFF 0x3f330000
float my_logf_nofma(float x){
  uint32_t ix=asuint(x);
  if (ix==0x3f800000) return 0;
  if (ix-0x00800000 >= 0x7f800000-0x00800000){ /* subnormal etc */
    if (ix*2==0) return -INFINITY; if (ix==0x7f800000) return x; if ((ix&0x80000000)||ix*2>=0xff000000) return NAN;
    ix=asuint(x*0x1p23f); ix-=23<<23; }
  uint32_t tmp=ix-OFF; int i=(tmp>>(23-4))%16; int k=(int32_t)tmp>>23; uint32_t iz=ix-(tmp&(0x1ffu<<23));
  double invc=T[i].invc, logc=T[i].logc, z=(double)asfloat(iz);
  double r=z*invc-1; double y0=logc+(double)k*Ln2; double r2=r*r; double y=A[1]*r+A[2]; y=A[0]*r2+y; y=y*r2+(y0+r);
  return (float)y;
}
float my_logf_fma(float x){
  uint32_t ix=asuint(x);
  if (ix==0x3f800000) return 0;
  if (ix-0x00800000 >= 0x7f800000-0x00800000){
    if (ix*2==0) return -INFINITY; if (ix==0x7f800000) return x; if ((ix&0x80000000)||ix*2>=0xff000000) return NAN;
    ix=asuint(x*0x1p23f); ix-=23<<23; }
  uint32_t tmp=ix-OFF; int i=(tmp>>(23-4))%16; int k=(int32_t)tmp>>23; uint32_t iz=ix-(tmp&(0x1ffu<<23));
  double invc=T[i].invc, logc=T[i].logc, z=(double)asfloat(iz);
  double r=fma(z,invc,-1.0); double y0=fma((double)k,Ln2,logc); double r2=r*r; double y=fma(A[1],r,A[2]); y=fma(A[0],r2,y); y=fma(y,r2,(y0+r));
  return (float)y;
}
static const float two25=3.3554432000e+07f, ivln10=4.3429449201e-01f, log10_2hi=3.0102920532e-01f, log10_2lo=7.9034151668e-07f;
float my_log10f(float x, float (*lg)(float)){
  int32_t hx=(int32_t)asuint(x),i,k=0; float y,z;
  if (hx<0x00800000){ if((hx&0x7fffffff)==0) return -INFINITY; if(hx<0) return NAN; k-=25; x*=two25; hx=(int32_t)asuint(x);}
  if (hx>=0x7f800000) return x+x;
  k+=(hx>>23)-127; i=((uint32_t)k&0x80000000u)>>31; hx=(hx&0x007fffff)|((0x7f-i)<<23); y=(float)(k+i); x=asfloat((uint32_t)hx);
  z=y*log10_2lo+ivln10*lg(x);
  return z+y*log10_2hi;
}
int main(int argc,char**argv){
  uint32_t stride = argc>1 ? (uint32_t)atoi(argv[1]) : 1;
  uint64_t bad_nf=0,bad_f=0,bad10_nf=0,bad10_f=0,n=0;
  for (uint32_t u=0x00800000; u<0x7f800000; u+=stride){
    float x=asfloat(u); float r=logf(x);
    if (asuint(r)!=asuint(my_logf_nofma(x))) bad_nf++;
    if (asuint(r)!=asuint(my_logf_fma(x))) bad_f++;
    float r10=log10f(x);
    if (asuint(r10)!=asuint(my_log10f(x,my_logf_nofma))) bad10_nf++;
    if (asuint(r10)!=asuint(my_log10f(x,my_logf_fma))) bad10_f++;
    n++;
  }
  printf("n=%lu logf mismatches: nofma=%lu fma=%lu ; log10f mismatches: nofma=%lu fma=%lu\n",n,bad_nf,bad_f,bad10_nf,bad10_f);
  return (bad_nf||bad10_nf) ? 1 : 0;
}
