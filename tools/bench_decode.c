#include "host_io.h"
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
static double now(void){struct timespec t; clock_gettime(CLOCK_MONOTONIC,&t); return t.tv_sec+t.tv_nsec*1e-9;}
int main(int argc,char**argv){
  fastf_lists_t L; if (fastf_lists_load(argv[2], argv[3], 1.0f, 926, &L)) { fprintf(stderr,"%s\n", fastf_last_error()); return 1; }
  int thr = argc>4? atoi(argv[4]):0;
  size_t cap = 1<<22; uint64_t*cb=malloc(cap*8),*gx=malloc(cap*8); uint32_t*um=malloc(cap*4),*me=malloc(cap*4);
  double t0=now(); fastf_bam_t*b=fastf_bam_open(argv[1],thr); if(!b){fprintf(stderr,"%s\n",fastf_last_error());return 1;}
  size_t tot=0; uint64_t acc=0; long n;
  while((n=fastf_bam_read_batch(b,L.cell_dict,L.feat_dict,cb,gx,um,me,cap))>0){ tot+=n; for(long i=0;i<n;i+=997) acc+=cb[i]^gx[i]^um[i]^me[i]; }
  double dt=now()-t0; printf("%zu records in %.3f s = %.2f M rec/s (threads=%d) chk=%llx\n",tot,dt,tot/dt/1e6,thr,(unsigned long long)acc);
  fastf_bam_close(b); return 0; }
/* stand-alone build: the two error hooks normally provided by umi_engine.hip */
static char g_e[512];
void fastf_set_error_(const char *m) { snprintf(g_e, sizeof g_e, "%s", m); }
const char *fastf_last_error(void) { return g_e; }
