// does the packed-fp32 activation producer (h2_act4, inline-asm v_pk_*_f32) give the scalar form's bits?  tools only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float h2_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2_f2 h2_pk_add(h2_f2 x, h2_f2 y) { h2_f2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ h2_f2 h2_pk_mul(h2_f2 x, h2_f2 y) { h2_f2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ h2_f2 h2_pk_sub(h2_f2 x, h2_f2 y) { h2_f2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y)); return r; }

__device__ __forceinline__ float silu_fast(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
__device__ __forceinline__ void split2h(float xs, unsigned short& h1, unsigned short& h2) {
    const _Float16 a = (_Float16)xs; const _Float16 b = (_Float16)(xs - (float)a);
    h1 = __builtin_bit_cast(unsigned short, a); h2 = __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ void h2_act4_hazard(const float4& a, const float4& b, int e, uint2& hi, uint2& lo) {
    const h2_f2 z0 = h2_pk_add(h2_f2{a.x, a.y}, h2_f2{b.x, b.y}), z1 = h2_pk_add(h2_f2{a.z, a.w}, h2_f2{b.z, b.w});
    const h2_f2 k = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.0f, 1.0f};
    const h2_f2 t0 = h2_pk_mul(z0, k), t1 = h2_pk_mul(z1, k);
    const h2_f2 d0 = h2_pk_add(h2_f2{__builtin_amdgcn_exp2f(t0.x), __builtin_amdgcn_exp2f(t0.y)}, one);
    const h2_f2 d1 = h2_pk_add(h2_f2{__builtin_amdgcn_exp2f(t1.x), __builtin_amdgcn_exp2f(t1.y)}, one);
    const h2_f2 h0 = h2_pk_mul(z0, h2_f2{__builtin_amdgcn_rcpf(d0.x), __builtin_amdgcn_rcpf(d0.y)});
    const h2_f2 h1 = h2_pk_mul(z1, h2_f2{__builtin_amdgcn_rcpf(d1.x), __builtin_amdgcn_rcpf(d1.y)});
    const h2_f2 s0 = {ldexpf(h0.x, e), ldexpf(h0.y, e)}, s1 = {ldexpf(h1.x, e), ldexpf(h1.y, e)};
    typedef _Float16 h2_h2 __attribute__((ext_vector_type(2)));
    const h2_h2 a0 = __builtin_convertvector(s0, h2_h2), a1 = __builtin_convertvector(s1, h2_h2);          // v_cvt_pk_f16_f32 (round to nearest even)
    const h2_f2 r0 = h2_pk_sub(s0, __builtin_convertvector(a0, h2_f2)), r1 = h2_pk_sub(s1, __builtin_convertvector(a1, h2_f2));
    const h2_h2 b0 = __builtin_convertvector(r0, h2_h2), b1 = __builtin_convertvector(r1, h2_h2);
    hi = make_uint2(__builtin_bit_cast(unsigned int, a0), __builtin_bit_cast(unsigned int, a1));
    lo = make_uint2(__builtin_bit_cast(unsigned int, b0), __builtin_bit_cast(unsigned int, b1));
}
__device__ __forceinline__ h2_f2 h2_pk_add_t(h2_f2 x, h2_f2 y) { h2_f2 r; asm("s_nop 0\n\tv_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ h2_f2 h2_pk_mul_t(h2_f2 x, h2_f2 y) { h2_f2 r; asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ void h2_act4(const float4& a, const float4& b, int e, uint2& hi, uint2& lo) {
    const h2_f2 z0 = h2_pk_add(h2_f2{a.x, a.y}, h2_f2{b.x, b.y}), z1 = h2_pk_add(h2_f2{a.z, a.w}, h2_f2{b.z, b.w});
    const h2_f2 k = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.0f, 1.0f};
    const h2_f2 t0 = h2_pk_mul(z0, k), t1 = h2_pk_mul(z1, k);
    const h2_f2 d0 = h2_pk_add_t(h2_f2{__builtin_amdgcn_exp2f(t0.x), __builtin_amdgcn_exp2f(t0.y)}, one);
    const h2_f2 d1 = h2_pk_add_t(h2_f2{__builtin_amdgcn_exp2f(t1.x), __builtin_amdgcn_exp2f(t1.y)}, one);
    const h2_f2 h0 = h2_pk_mul_t(z0, h2_f2{__builtin_amdgcn_rcpf(d0.x), __builtin_amdgcn_rcpf(d0.y)});
    const h2_f2 h1 = h2_pk_mul_t(z1, h2_f2{__builtin_amdgcn_rcpf(d1.x), __builtin_amdgcn_rcpf(d1.y)});
    const h2_f2 s0 = {ldexpf(h0.x, e), ldexpf(h0.y, e)}, s1 = {ldexpf(h1.x, e), ldexpf(h1.y, e)};
    typedef _Float16 h2_h2 __attribute__((ext_vector_type(2)));
    const h2_h2 a0 = __builtin_convertvector(s0, h2_h2), a1 = __builtin_convertvector(s1, h2_h2);          // v_cvt_pk_f16_f32 (round to nearest even)
    const h2_f2 r0 = h2_pk_sub(s0, __builtin_convertvector(a0, h2_f2)), r1 = h2_pk_sub(s1, __builtin_convertvector(a1, h2_f2));
    const h2_h2 b0 = __builtin_convertvector(r0, h2_h2), b1 = __builtin_convertvector(r1, h2_h2);
    hi = make_uint2(__builtin_bit_cast(unsigned int, a0), __builtin_bit_cast(unsigned int, a1));
    lo = make_uint2(__builtin_bit_cast(unsigned int, b0), __builtin_bit_cast(unsigned int, b1));
}

__global__ void k_act(const float4* a, const float4* b, uint2* hi, uint2* lo, uint2* hi_s, uint2* lo_s) {
    const int i = threadIdx.x;
    h2_act4(a[i], b[i], 3, hi[i], lo[i]);
    const float h[4] = {silu_fast(a[i].x + b[i].x), silu_fast(a[i].y + b[i].y), silu_fast(a[i].z + b[i].z), silu_fast(a[i].w + b[i].w)};
    unsigned short p1[4], p2[4];
    for (int e = 0; e < 4; ++e) split2h(ldexpf(h[e], 3), p1[e], p2[e]);
    hi_s[i] = make_uint2(p1[0] | (unsigned)p1[1] << 16, p1[2] | (unsigned)p1[3] << 16);
    lo_s[i] = make_uint2(p2[0] | (unsigned)p2[1] << 16, p2[2] | (unsigned)p2[3] << 16);
}
__global__ void k_act_hazard(const float4* a, const float4* b, uint2* hi, uint2* lo, uint2* hi_s, uint2* lo_s) {
    const int i = threadIdx.x;
    h2_act4_hazard(a[i], b[i], 3, hi[i], lo[i]);
    const float h[4] = {silu_fast(a[i].x + b[i].x), silu_fast(a[i].y + b[i].y), silu_fast(a[i].z + b[i].z), silu_fast(a[i].w + b[i].w)};
    unsigned short p1[4], p2[4];
    for (int e = 0; e < 4; ++e) split2h(ldexpf(h[e], 3), p1[e], p2[e]);
    hi_s[i] = make_uint2(p1[0] | (unsigned)p1[1] << 16, p1[2] | (unsigned)p1[3] << 16);
    lo_s[i] = make_uint2(p2[0] | (unsigned)p2[1] << 16, p2[2] | (unsigned)p2[3] << 16);
}
__global__ void k(const float* a, const float* b, float* out) {
    const int i = threadIdx.x;
    h2_f2 x = {a[2 * i], a[2 * i + 1]}, y = {b[2 * i], b[2 * i + 1]};
    h2_f2 s = h2_pk_add(x, y), m = h2_pk_mul(x, y), d = h2_pk_sub(x, y);
    out[6 * i] = s.x; out[6 * i + 1] = s.y; out[6 * i + 2] = m.x; out[6 * i + 3] = m.y; out[6 * i + 4] = d.x; out[6 * i + 5] = d.y;
}
int main() {
    float ha[128], hb[128], ho[384];
    for (int i = 0; i < 128; ++i) { ha[i] = 0.37f * i - 11.0f; hb[i] = 1.0f / (1 + i); }
    float *a, *b, *o;
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&o, 1536);
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o);
    hipMemcpy(ho, o, 1536, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i)
        for (int h = 0; h < 2; ++h) {
            const float x = ha[2 * i + h], y = hb[2 * i + h];
            if (ho[6 * i + h] != x + y || ho[6 * i + 2 + h] != x * y || ho[6 * i + 4 + h] != x - y) {
                if (bad < 6) printf("lane %d half %d: x %g y %g  add %g (%g)  mul %g (%g)  sub %g (%g)\n", i, h, x, y, ho[6 * i + h], x + y, ho[6 * i + 2 + h], x * y, ho[6 * i + 4 + h], x - y);
                ++bad;
            }
        }
    printf("mismatches: %d of 384\n", bad);
for (int form = 0; form < 2; ++form)
    {
        float4 *da, *db; uint2 *o4; float h4a[256], h4b[256]; uint2 r[256];
        for (int i = 0; i < 256; ++i) { h4a[i] = 0.11f * i - 13.0f; h4b[i] = 3.0f / (1 + i) - 1.0f; }
        hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&o4, 2048);
        hipMemcpy(da, h4a, 1024, hipMemcpyHostToDevice); hipMemcpy(db, h4b, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(form ? k_act_hazard : k_act, dim3(1), dim3(64), 0, 0, da, db, o4, o4 + 64, o4 + 128, o4 + 192);
        hipMemcpy(r, o4, 2048, hipMemcpyDeviceToHost);
        int bad2 = 0;
        for (int i = 0; i < 128; ++i) if (r[i].x != r[128 + i].x || r[i].y != r[128 + i].y) { if (bad2 < 4) printf("act %d: %08x %08x vs %08x %08x\n", i, r[i].x, r[i].y, r[128 + i].x, r[128 + i].y); ++bad2; }
        printf("activation mismatches (%s): %d of 128\n", form ? "no wait state after v_rcp / v_exp" : "product form", bad2);
    }
    return 0;
}
