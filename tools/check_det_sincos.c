#include <math.h>
#include <stdio.h>
#include <stdlib.h>
static void det_sincos(double x, double* s, double* c) {
  double q = rint(x * 0.63661977236758134308);
  long long qi = (long long)q;
  double r = fma(q, -1.57079632673412561417e+00, x);
  r = fma(q, -6.07710050650619224932e-11, r);
  double z = r * r;
  double ps = 1.58969099521155010221e-10;
  ps = fma(ps, z, -2.50507602534068634195e-08);
  ps = fma(ps, z, 2.75573137070700676789e-06);
  ps = fma(ps, z, -1.98412698298579493134e-04);
  ps = fma(ps, z, 8.33333333332248946124e-03);
  ps = fma(ps, z, -1.66666666666666324348e-01);
  double sn = fma(ps * z, r, r);
  double pc = -1.13596475577881948265e-11;
  pc = fma(pc, z, 2.08757232129817482790e-09);
  pc = fma(pc, z, -2.75573143513906633035e-07);
  pc = fma(pc, z, 2.48015872894767294178e-05);
  pc = fma(pc, z, -1.38888888888741095749e-03);
  pc = fma(pc, z, 4.16666666666666019037e-02);
  double cs = fma(pc * z, z, fma(-0.5, z, 1.0));
  double ss = (qi & 1) ? cs : sn, cc = (qi & 1) ? sn : cs;
  if (qi & 2) ss = -ss;
  if ((qi + 1) & 2) cc = -cc;
  *s = ss; *c = cc;
}
int main(){ double maxs=0,maxc=0; srand(1);
  for (int i=0;i<20000000;i++){ double x = ((double)rand()/RAND_MAX*2-1)*(i%2?100.0:0.3); double s,c; det_sincos(x,&s,&c);
    long double ts=sinl((long double)x), tc=cosl((long double)x);
    double us = fabs((double)((s-ts)/ (long double)(nextafter(fabs((double)ts),INFINITY)-fabs((double)ts))));
    double uc = fabs((double)((c-tc)/ (long double)(nextafter(fabs((double)tc),INFINITY)-fabs((double)tc))));
    if(us>maxs)maxs=us; if(uc>maxc)maxc=uc; }
  printf("max ulp err sin %.3f cos %.3f\n",maxs,maxc); double s,c; det_sincos(-0.0,&s,&c); printf("%g %g\n",s,c); return 0; }
