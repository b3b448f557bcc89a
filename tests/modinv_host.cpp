// host build of the engine's modular inversion (halo2ecc_s_amd/csrc/modinv62.h) for tests/test_modinv_cpu.py
#include "../halo2ecc_s_amd/csrc/modinv62.h"
extern "C" void modinv_4(const uint64_t* a, const uint64_t* p, uint64_t* out, int n) {
    for (int i = 0; i < n; i++) modinv62::inv<4>(a + 4 * i, p, out + 4 * i);
}
extern "C" void modinv_6(const uint64_t* a, const uint64_t* p, uint64_t* out, int n) {
    for (int i = 0; i < n; i++) modinv62::inv<6>(a + 6 * i, p, out + 6 * i);
}
