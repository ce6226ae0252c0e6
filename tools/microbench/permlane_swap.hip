// semantics check of gfx950's v_permlane32_swap / v_permlane16_swap as pose_opt.hip's swap_add uses them:
// prints PASS when, for a = lane and b = 100 + lane,
//   permlane32_swap: lanes < 32 get (a_own, a_of_lane+32), lanes >= 32 get (b_of_lane-32, b_own)
//   permlane16_swap: even 16-lane rows get (a_own, a_of_lane+16), odd rows get (b_of_lane-16, b_own)
// hipcc --offload-arch=gfx950 -O2 permlane_swap.hip -o permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *out)
{
    const unsigned l = threadIdx.x, a = l, b = 100 + l;
    u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = r[0]; out[64 + l] = r[1];
    u2 q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (unsigned l = 0; l < 64; ++l) {
        const unsigned e0 = l < 32 ? l : 100 + l - 32, e1 = l < 32 ? l + 32 : 100 + l;
        bad += h[l] != e0 || h[64 + l] != e1;
        const bool odd = (l >> 4) & 1;
        const unsigned f0 = !odd ? l : 100 + l - 16, f1 = !odd ? l + 16 : 100 + l;
        bad += h[128 + l] != f0 || h[192 + l] != f1;
    }
    printf(bad ? "FAIL %d\n" : "PASS\n", bad);
    if (bad) for (int l = 0; l < 64; ++l) printf("%2d: s32 (%3u %3u) s16 (%3u %3u)\n", l, h[l], h[64 + l], h[128 + l], h[192 + l]);
    return bad != 0;
}
