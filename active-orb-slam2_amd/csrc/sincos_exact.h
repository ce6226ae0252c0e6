// Deterministic float sin/cos of a float radian angle in [0, 2*pi+]: evaluated in IEEE double
// with a fixed sequence of +,-,* (no FMA contraction: build with -ffp-contract=off), then rounded
// to float.  Used for the rBRIEF steering a = cos(angle), b = sin(angle)
// (reference src/ORBextractor.cc:112-113, which calls libm cosf/sinf whose last bit is
// glibc-version dependent).  The same operation sequence runs on host and device, so both sides
// agree bit for bit; tests pin it against libm on a dense sample.
#pragma once

#if defined(__HIPCC__)
#define AOS2_HD __host__ __device__ inline
#else
#define AOS2_HD inline
#endif

namespace aos2 {

AOS2_HD void sincos_exact(float angle_rad, float *s_out, float *c_out)
{
    const double x = (double)angle_rad;
    // k = round(x * 2/pi), x >= 0
    const int k = (int)(x * 6.36619772367581382433e-01 + 0.5);
    const double fk = (double)k;
    // Cody-Waite: pi/2 = PIO2_1 + PIO2_1T (first 33 bits + tail)
    double r = x - fk * 1.57079632673412561417e+00;
    r = r - fk * 6.07710050650619224932e-11;
    const double z = r * r;
    // fdlibm __kernel_sin / __kernel_cos minimax coefficients on [-pi/4, pi/4]
    double ps = 1.58969099521155010221e-10;
    ps = ps * z + -2.50507602534068634195e-08;
    ps = ps * z + 2.75573137070700676789e-06;
    ps = ps * z + -1.98412698298579493134e-04;
    ps = ps * z + 8.33333333332248946124e-03;
    ps = ps * z + -1.66666666666666324348e-01;
    const double sn = r + (r * z) * ps;
    double pc = -1.13596475577881948265e-11;
    pc = pc * z + 2.08757232129817482790e-09;
    pc = pc * z + -2.75573143513906633035e-07;
    pc = pc * z + 2.48015872894767294178e-05;
    pc = pc * z + -1.38888888888741095749e-03;
    pc = pc * z + 4.16666666666666019037e-02;
    const double cs = (1.0 - 0.5 * z) + (z * z) * pc;
    double s, c;
    switch (k & 3) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

}  // namespace aos2
