// Exhaustive host check behind div3() of simple-path-tracer_amd/csrc/hip/bezier.h: x / 3.0f against the two-fma form for all 2^32 bit patterns.
// gcc -O2 -ffp-contract=off -mfma -o /tmp/check_div3 tools/check_div3.c -lpthread -lm && /tmp/check_div3   (16 s on 8 cores; expect 3 mismatches: +-inf, -0)
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
static inline float div3(float x){ const float r=0x1.555556p-2f; float q=x*r; float e=fmaf(-3.0f,q,x); return fmaf(e,r,q); }
static uint64_t bad[8]; static uint32_t firstbad[8];
void* run(void* a){ int t=(int)(intptr_t)a; uint64_t b=0; for(uint64_t i=(uint64_t)t<<29;i<((uint64_t)(t+1)<<29);++i){ uint32_t u=(uint32_t)i; float x; memcpy(&x,&u,4); if(x!=x) continue; float a1=x/3.0f, a2=div3(x); uint32_t p,q; memcpy(&p,&a1,4); memcpy(&q,&a2,4); if(p!=q){ if(!b) firstbad[t]=u; ++b; } } bad[t]=b; return 0; }
int main(){ pthread_t th[8]; for(int t=0;t<8;++t) pthread_create(&th[t],0,run,(void*)(intptr_t)t); uint64_t tot=0; for(int t=0;t<8;++t){ pthread_join(th[t],0); tot+=bad[t]; if(bad[t]) printf("t%d bad %llu first %08x\n",t,(unsigned long long)bad[t],firstbad[t]); } printf("total mismatches %llu\n",(unsigned long long)tot); return 0; }
