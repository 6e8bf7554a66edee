// Timing/accuracy check of the host symmetric eigensolver (K7).  gcc -O2 -I include tools/eig_bench.c -Lgcge_amd/lib -lgcge_host -Wl,-rpath,$PWD/gcge_amd/lib -lm -o /tmp/eig_bench
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <math.h>
int GCGE_SymEig(char uplo, int n, const double *a, int lda, double *w, double *z, int ldz, double *work);
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+1e-9*t.tv_nsec;}
int main(int argc,char**argv){
  int n=argc>1?atoi(argv[1]):512; double*a=malloc(sizeof(double)*n*n),*z=malloc(sizeof(double)*n*n),*w=malloc(8*n),*wk=malloc(8*4*n);
  srand(1); for(int j=0;j<n;++j)for(int i=0;i<=j;++i){double v=rand()/(double)RAND_MAX-0.5; a[(size_t)j*n+i]=v; a[(size_t)i*n+j]=v;}
  for(int i=0;i<n;++i)a[(size_t)i*n+i]+=i*0.01;
  int info; double t; for(int rep=0;rep<4;++rep){ t=now(); info=GCGE_SymEig('U',n,a,n,w,z,n,wk); t=now()-t; printf("rep %.3f\n",t);}
  /* residual check */
  double maxr=0, maxo=0;
  for(int j=0;j<n;j+=37){ for(int i=0;i<n;++i){double s=0; for(int k=0;k<n;++k)s+=a[(size_t)k*n+i]*z[(size_t)j*n+k]; s-=w[j]*z[(size_t)j*n+i]; if(fabs(s)>maxr)maxr=fabs(s);} 
    for(int j2=0;j2<n;j2+=41){double s=0; for(int k=0;k<n;++k)s+=z[(size_t)j*n+k]*z[(size_t)j2*n+k]; if(j==j2)s-=1; if(fabs(s)>maxo)maxo=fabs(s);} }
  printf("n=%d info=%d time %.3f s  max resid %.2e  max orth %.2e  w0 %.12f wlast %.12f\n",n,info,t,maxr,maxo,w[0],w[n-1]);
  return 0;}
