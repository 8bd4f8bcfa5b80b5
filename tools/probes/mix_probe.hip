#include <hip/hip_runtime.h>
#include <cstring>
#include <cmath>
#include <cstdio>
static float h2f_host(unsigned short x) { unsigned s = (x >> 15) & 1, e = (x >> 10) & 31, m = x & 1023; float v; if (e == 0) v = ldexpf((float)m, -24); else if (e == 31) v = INFINITY; else v = ldexpf((float)(m | 1024), (int)e - 25); return s ? -v : v; }
__global__ void k(const float* in, unsigned* out) {
    float a = in[threadIdx.x], b = in[threadIdx.x + 64];
    unsigned hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    float la, lb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
    unsigned lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(la, lb));
    out[threadIdx.x] = hi;
    out[threadIdx.x + 64] = lo;
}
int main() {
    float *in; unsigned *out;
    hipMalloc(&in, 128 * 4); hipMalloc(&out, 128 * 4);
    float h[128]; for (int i = 0; i < 128; ++i) h[i] = 0.37f * (i - 40) + 1e-3f * i * i;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out);
    unsigned o[128]; hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        // reference: hi = trunc to fp16 (RTZ), lo = RTZ(a - hi)
        auto h2f = h2f_host;
        float a = h[i], b = h[i + 64];
        float ha = h2f(o[i] & 0xffff), hb = h2f(o[i] >> 16), la = h2f(o[i + 64] & 0xffff), lb = h2f(o[i + 64] >> 16);
        double ea = (double)a - ha - la, eb = (double)b - hb - lb;
        if (fabs(ea) > fabs(a) * 1e-6 + 1e-9 || fabs(eb) > fabs(b) * 1e-6 + 1e-9) { ++bad; printf("%d: a %g ha %g la %g err %g | b %g hb %g lb %g err %g\n", i, a, ha, la, ea, b, hb, lb, eb); }
    }
    printf("bad %d\n", bad);
    return 0;
}
