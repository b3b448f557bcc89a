// host build of the engine's modular inversion (halo2ecc_s_amd/csrc/modinv62.h) for tests/test_modinv_cpu.py
#include "../halo2ecc_s_amd/csrc/modinv62.h"
extern "C" void modinv_4(const uint64_t* a, const uint64_t* p, uint64_t* out, int n) {
    for (int i = 0; i < n; i++) modinv62::inv<4>(a + 4 * i, p, out + 4 * i);
}
extern "C" void modinv_6(const uint64_t* a, const uint64_t* p, uint64_t* out, int n) {
    for (int i = 0; i < n; i++) modinv62::inv<6>(a + 6 * i, p, out + 6 * i);
}
// the memory-resident form (state in caller-provided memory: LDS on the device)
template <int N>
static void inv_mem_n(const uint64_t* a, const uint64_t* p, uint64_t* out) {
    constexpr int NL = (64 * N + 61) / 62;
    int64_t st[4 * NL];
    uint64_t pw[N + 1];
    for (int i = 0; i < N; i++) pw[i] = p[i];
    pw[N] = 0;
    modinv62::inv_mem<N, int64_t*, const uint64_t*>(a, pw, out, st, st + NL, st + 2 * NL, st + 3 * NL);
}
extern "C" void modinv_mem_4(const uint64_t* a, const uint64_t* p, uint64_t* out, int n) {
    for (int i = 0; i < n; i++) inv_mem_n<4>(a + 4 * i, p, out + 4 * i);
}
extern "C" void modinv_mem_6(const uint64_t* a, const uint64_t* p, uint64_t* out, int n) {
    for (int i = 0; i < n; i++) inv_mem_n<6>(a + 6 * i, p, out + 6 * i);
}
