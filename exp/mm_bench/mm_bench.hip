// micro-benchmark + equality check: product-scanning (FIPS) Montgomery multiplication with a 96-bit column accumulator
// (v_mad_u64_u32 + v_addc per product) against the operand-scanning CIOS form.  hipcc --offload-arch=gfx950 -O3 mm_bench.hip -o mm_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint64_t u64; typedef uint32_t u32;
template <int N> struct Wd { u64 v[N]; };
template <int N> __device__ __forceinline__ u32 limb32(const Wd<N>& a, int i) { return (u32)(a.v[i >> 1] >> ((i & 1) * 32)); }

// FIPS / product scanning, 32-bit limbs, 96-bit column accumulator (lo64 + carry word)
struct Acc { u64 lo; u32 hi; };
__device__ __forceinline__ void mac(Acc& A, u32 a, u32 b) {
#if 1
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(A.lo), "+v"(A.hi) : "v"(a), "v"(b) : "vcc");
#else
    u64 p = (u64)a * b;
    A.lo += p;
    A.hi += A.lo < p;
#endif
}
template <int N>
__device__ __forceinline__ Wd<N> mont_mul_fips(const Wd<N>& a, const Wd<N>& b, const Wd<N>& p, u32 minv) {
    constexpr int L = 2 * N;
    u32 m[L], r[L];
    Acc A{0, 0};
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mac(A, limb32<N>(a, i), limb32<N>(b, k - i));
#pragma unroll
        for (int i = 0; i < k; i++) mac(A, m[i], limb32<N>(p, k - i));
        m[k] = (u32)A.lo * minv;
        mac(A, m[k], limb32<N>(p, 0));
        A.lo = (A.lo >> 32) | ((u64)A.hi << 32);
        A.hi = 0;
    }
#pragma unroll
    for (int k = L; k < 2 * L; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) mac(A, limb32<N>(a, i), limb32<N>(b, k - i));
#pragma unroll
        for (int i = k - L + 1; i < L; i++) mac(A, m[i], limb32<N>(p, k - i));
        r[k - L] = (u32)A.lo;
        A.lo = (A.lo >> 32) | ((u64)A.hi << 32);
        A.hi = 0;
    }
    Wd<N> o;
#pragma unroll
    for (int i = 0; i < N; i++) o.v[i] = (u64)r[2 * i] | ((u64)r[2 * i + 1] << 32);
    // result < 2p: conditional subtract
    bool ge = (u32)A.lo != 0;
    if (!ge) {
        ge = true;
#pragma unroll
        for (int i = N - 1; i >= 0; i--) {
            if (o.v[i] != p.v[i]) { ge = o.v[i] > p.v[i]; break; }
        }
    }
    if (ge) {
        u64 bor = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            u64 d = o.v[i] - p.v[i], d2 = d - bor;
            bor = (o.v[i] < p.v[i]) | (d < bor);
            o.v[i] = d2;
        }
    }
    return o;
}

// the engine's current CIOS multiplication (engine.hip mont_mul) for comparison
template <int N>
__device__ __forceinline__ Wd<N> mont_mul_cios(const Wd<N>& a, const Wd<N>& b, const Wd<N>& p, u32 minv) {
    constexpr int L32 = 2 * N;
    u32 t[L32 + 2];
#pragma unroll
    for (int i = 0; i < L32 + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < L32; i++) {
        u32 bi = limb32<N>(b, i);
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < L32; j++) {
            u64 s = (u64)limb32<N>(a, j) * bi + t[j] + c;
            t[j] = (u32)s;
            c = s >> 32;
        }
        u64 s = (u64)t[L32] + c;
        t[L32] = (u32)s;
        t[L32 + 1] = (u32)(s >> 32);
        u32 m = t[0] * minv;
        c = ((u64)m * limb32<N>(p, 0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < L32; j++) {
            u64 s2 = (u64)m * limb32<N>(p, j) + t[j] + c;
            t[j - 1] = (u32)s2;
            c = s2 >> 32;
        }
        s = (u64)t[L32] + c;
        t[L32 - 1] = (u32)s;
        t[L32] = t[L32 + 1] + (u32)(s >> 32);
    }
    Wd<N> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = (u64)t[2 * i] | ((u64)t[2 * i + 1] << 32);
    bool ge = t[L32] != 0;
    if (!ge) {
        ge = true;
#pragma unroll
        for (int i = N - 1; i >= 0; i--) {
            if (r.v[i] != p.v[i]) { ge = r.v[i] > p.v[i]; break; }
        }
    }
    if (ge) {
        u64 bor = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            u64 d = r.v[i] - p.v[i], d2 = d - bor;
            bor = (r.v[i] < p.v[i]) | (d < bor);
            r.v[i] = d2;
        }
    }
    return r;
}
template <int N, int WHICH>
__global__ void k(const Wd<N>* a, const Wd<N>* b, Wd<N>* out, Wd<N> p, u32 minv, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Wd<N> x = a[i], y = b[i];
    for (int t = 0; t < iters; t++) x = WHICH ? mont_mul_fips<N>(x, y, p, minv) : mont_mul_cios<N>(x, y, p, minv);
    out[i] = x;
}
#include <stdio.h>
#include <stdlib.h>
#include <vector>
template <int N>
int run(const char* name, const u64* pw) {
    const int n = 256 * 1024;
    Wd<N> p;
    for (int i = 0; i < N; i++) p.v[i] = pw[i];
    u32 p0 = (u32)p.v[0], inv = p0;
    for (int i = 0; i < 5; i++) inv *= 2 - p0 * inv;
    u32 minv = (u32)(0u - inv);
    std::vector<Wd<N>> ha(n), hb(n), o0(n), o1(n);
    srand(1);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < N; j++) {
            u64 x = 0, y = 0;
            for (int k = 0; k < 4; k++) { x = (x << 16) ^ (u64)rand(); y = (y << 16) ^ (u64)rand(); }
            ha[i].v[j] = j == N - 1 ? x % (pw[N - 1]) : x;   // below p (top word below p's top word)
            hb[i].v[j] = j == N - 1 ? y % (pw[N - 1]) : y;
        }
    for (int j = 0; j < N; j++) { ha[0].v[j] = 0; hb[1].v[j] = 0; ha[2].v[j] = pw[j]; hb[2].v[j] = pw[j]; ha[3].v[j] = pw[j]; }
    ha[2].v[0] -= 1; hb[2].v[0] -= 1; ha[3].v[0] -= 1;   // p - 1
    Wd<N>*da, *db, *dout;
    hipMalloc(&da, n * sizeof(Wd<N>)); hipMalloc(&db, n * sizeof(Wd<N>)); hipMalloc(&dout, n * sizeof(Wd<N>));
    hipMemcpy(da, ha.data(), n * sizeof(Wd<N>), hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), n * sizeof(Wd<N>), hipMemcpyHostToDevice);
    int bad = 0;
    for (int iters : {1, 3}) {
        k<N, 0><<<n / 64, 64>>>(da, db, dout, p, minv, iters);
        hipMemcpy(o0.data(), dout, n * sizeof(Wd<N>), hipMemcpyDeviceToHost);
        k<N, 1><<<n / 64, 64>>>(da, db, dout, p, minv, iters);
        hipMemcpy(o1.data(), dout, n * sizeof(Wd<N>), hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < N; j++) bad += o0[i].v[j] != o1[i].v[j];
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (which) k<N, 1><<<n / 64, 64>>>(da, db, dout, p, minv, 2000); else k<N, 0><<<n / 64, 64>>>(da, db, dout, p, minv, 2000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[which], e0, e1);
        }
    }
    printf("%s: mismatching words %d; cios %.2f ms, fips %.2f ms for %d x 2000 multiplications (%.1f / %.1f G mul/s)\n", name, bad, ms[0], ms[1], n,
           n * 2000.0 / ms[0] / 1e6, n * 2000.0 / ms[1] / 1e6);
    return bad;
}
int main() {
    const u64 bn_fq[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    const u64 bls_fq[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    int bad = run<4>("bn256 Fq", bn_fq);
    bad += run<6>("bls12_381 Fq", bls_fq);
    return bad != 0;
}
