// Host compile of the PRODUCT's trig header (gridworld_amd/csrc/igw_trig.h) for tests:
// lets the oracle replay exactly the sin/cos/atan2 the HIP kernels use ("device-trig" oracle mode),
// and lets tests/test_trig.py check the implementation against mpmath.  Test infrastructure only.
#include "../gridworld_amd/csrc/igw_trig.h"

extern "C" {
double igw_host_sin(double x) { double s, c; igw::igw_sincos(x, &s, &c); return s; }
double igw_host_cos(double x) { double s, c; igw::igw_sincos(x, &s, &c); return c; }
double igw_host_atan2(double y, double x) { return igw::igw_atan2(y, x); }
void igw_host_sincos_array(const double* x, double* s, double* c, long n) {
    for (long i = 0; i < n; i++) igw::igw_sincos(x[i], s + i, c + i);
}
// the two evaluations separately (tests/test_trig.py: acceptance rate and error of the quick one)
void igw_host_sincos_quick_array(const double* x, double* s, double* c, unsigned char* ok, long n) {
    for (long i = 0; i < n; i++) ok[i] = igw::igw_sincos_quick(x[i], s + i, c + i) ? 1 : 0;
}
void igw_host_sincos_accurate_array(const double* x, double* s, double* c, long n) {
    for (long i = 0; i < n; i++) igw::igw_sincos_accurate(x[i], s + i, c + i);
}
void igw_host_atan2_quick_array(const double* y, const double* x, double* out, unsigned char* ok, long n) {
    for (long i = 0; i < n; i++) ok[i] = igw::igw_atan2_quick(y[i], x[i], out + i) ? 1 : 0;
}
void igw_host_atan2_accurate_array(const double* y, const double* x, double* out, long n) {
    for (long i = 0; i < n; i++) out[i] = igw::igw_atan2_accurate(y[i], x[i]);
}
void igw_host_atan2_array(const double* y, const double* x, double* out, long n) {
    for (long i = 0; i < n; i++) out[i] = igw::igw_atan2(y[i], x[i]);
}
}
