// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// Fixed-capacity unsigned big integer standing in for num-bigint's `BigUint`, which the
// reference uses for every witness-side integer computation (dependency `num-bigint` 0.4.4 /
// `num-integer` 0.1.46, Cargo.lock:595-631; source not under /root/reference).  Only the
// operations the reference's hot path calls are provided: +, -, *, div_rem, shifts, masks,
// bit tests, comparison, lcm/gcd (range_info.rs pre_check), modular inverse / pow (stand-ins
// for `W::invert()` of the un-vendored field crates, integer_chip.rs:524-527).
#pragma once
#include <cstdint>
#include <cstring>
#include <cassert>
#include <string>
#include <stdexcept>
#include <algorithm>

namespace h2o {

typedef unsigned __int128 u128;

struct BigUint {
    static const int CAP = 28;  // 1792 bits: enough for lcm(n, 2^540) and 774-bit products
    uint64_t w[CAP];
    int n;  // number of significant limbs (w[n-1] != 0), 0 for zero

    BigUint() : n(0) { std::memset(w, 0, sizeof(w)); }
    BigUint(uint64_t v) : n(v ? 1 : 0) { std::memset(w, 0, sizeof(w)); w[0] = v; }

    static BigUint from_limbs(const uint64_t* p, int cnt) {
        BigUint r;
        assert(cnt <= CAP);
        for (int i = 0; i < cnt; i++) r.w[i] = p[i];
        r.n = cnt;
        r.trim();
        return r;
    }
    static BigUint from_bytes_le(const uint8_t* p, int cnt) {
        BigUint r;
        assert(cnt <= CAP * 8);
        for (int i = 0; i < cnt; i++) r.w[i / 8] |= (uint64_t)p[i] << (8 * (i % 8));
        r.n = (cnt + 7) / 8;
        r.trim();
        return r;
    }
    static BigUint from_hex(const char* s) {
        BigUint r;
        if (s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) s += 2;
        int len = (int)std::strlen(s);
        for (int i = 0; i < len; i++) {
            char c = s[len - 1 - i];
            if (c == '_') throw std::runtime_error("no separators");
            uint64_t v = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : c - 'A' + 10;
            assert(i / 16 < CAP);
            r.w[i / 16] |= v << (4 * (i % 16));
        }
        r.n = (len + 15) / 16;
        r.trim();
        return r;
    }
    static BigUint from_dec(const char* s) {
        BigUint r;
        for (; *s; s++) r = r * BigUint(10) + BigUint((uint64_t)(*s - '0'));
        return r;
    }
    void to_bytes_le(uint8_t* out, int cnt) const {
        for (int i = 0; i < cnt; i++) out[i] = (i / 8 < CAP) ? (uint8_t)(w[i / 8] >> (8 * (i % 8))) : 0;
    }
    std::string to_hex() const {
        if (n == 0) return "0";
        static const char* d = "0123456789abcdef";
        std::string s;
        bool lead = true;
        for (int i = n - 1; i >= 0; i--)
            for (int j = 15; j >= 0; j--) {
                int v = (w[i] >> (4 * j)) & 15;
                if (lead && v == 0) continue;
                lead = false;
                s.push_back(d[v]);
            }
        return s;
    }

    void trim() {
        while (n > 0 && w[n - 1] == 0) n--;
    }
    bool is_zero() const { return n == 0; }
    // number of bits, like BigUint::bits()
    uint64_t bits() const {
        if (n == 0) return 0;
        return (uint64_t)(n - 1) * 64 + (64 - __builtin_clzll(w[n - 1]));
    }
    bool bit(uint64_t i) const {
        if (i / 64 >= (uint64_t)CAP) return false;
        return (w[i / 64] >> (i % 64)) & 1;
    }
    uint64_t low_u64() const { return w[0]; }

    static int cmp(const BigUint& a, const BigUint& b) {
        if (a.n != b.n) return a.n < b.n ? -1 : 1;
        for (int i = a.n - 1; i >= 0; i--)
            if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
        return 0;
    }
    bool operator==(const BigUint& o) const { return cmp(*this, o) == 0; }
    bool operator!=(const BigUint& o) const { return cmp(*this, o) != 0; }
    bool operator<(const BigUint& o) const { return cmp(*this, o) < 0; }
    bool operator<=(const BigUint& o) const { return cmp(*this, o) <= 0; }
    bool operator>(const BigUint& o) const { return cmp(*this, o) > 0; }
    bool operator>=(const BigUint& o) const { return cmp(*this, o) >= 0; }

    BigUint operator+(const BigUint& o) const {
        BigUint r;
        int m = std::max(n, o.n);
        u128 c = 0;
        for (int i = 0; i < m; i++) {
            c += (u128)w[i] + o.w[i];
            r.w[i] = (uint64_t)c;
            c >>= 64;
        }
        r.n = m;
        if (c) {
            assert(m < CAP);
            r.w[m] = (uint64_t)c;
            r.n = m + 1;
        }
        return r;
    }
    // panics on underflow like BigUint's Sub
    BigUint operator-(const BigUint& o) const {
        if (cmp(*this, o) < 0) throw std::runtime_error("BigUint subtraction underflow");
        BigUint r;
        uint64_t borrow = 0;
        for (int i = 0; i < n; i++) {
            u128 t = (u128)w[i] - o.w[i] - borrow;
            r.w[i] = (uint64_t)t;
            borrow = (uint64_t)(t >> 64) & 1;
        }
        r.n = n;
        r.trim();
        return r;
    }
    BigUint operator*(const BigUint& o) const {
        BigUint r;
        if (n == 0 || o.n == 0) return r;
        assert(n + o.n <= CAP);
        for (int i = 0; i < n; i++) {
            u128 c = 0;
            for (int j = 0; j < o.n; j++) {
                c += (u128)w[i] * o.w[j] + r.w[i + j];
                r.w[i + j] = (uint64_t)c;
                c >>= 64;
            }
            r.w[i + o.n] = (uint64_t)c;
        }
        r.n = n + o.n;
        r.trim();
        return r;
    }
    BigUint operator<<(uint64_t s) const {
        BigUint r;
        if (n == 0) return r;
        int ws = (int)(s / 64), bs = (int)(s % 64);
        assert(n + ws + 1 <= CAP);
        for (int i = n - 1; i >= 0; i--) {
            r.w[i + ws] |= w[i] << bs;
            if (bs) r.w[i + ws + 1] |= w[i] >> (64 - bs);
        }
        r.n = n + ws + 1;
        r.trim();
        return r;
    }
    BigUint operator>>(uint64_t s) const {
        BigUint r;
        int ws = (int)(s / 64), bs = (int)(s % 64);
        if (ws >= n) return r;
        for (int i = ws; i < n; i++) {
            r.w[i - ws] = w[i] >> bs;
            if (bs && i + 1 < n) r.w[i - ws] |= w[i + 1] << (64 - bs);
        }
        r.n = n - ws;
        r.trim();
        return r;
    }
    BigUint operator&(const BigUint& o) const {
        BigUint r;
        int m = std::min(n, o.n);
        for (int i = 0; i < m; i++) r.w[i] = w[i] & o.w[i];
        r.n = m;
        r.trim();
        return r;
    }

    // Knuth algorithm D. Returns (quotient, remainder) like num_integer::Integer::div_rem.
    static void div_rem(const BigUint& a, const BigUint& b, BigUint& q, BigUint& r) {
        if (b.n == 0) throw std::runtime_error("BigUint division by zero");
        if (cmp(a, b) < 0) {
            q = BigUint();
            r = a;
            return;
        }
        if (b.n == 1) {
            BigUint qq;
            u128 rem = 0;
            for (int i = a.n - 1; i >= 0; i--) {
                u128 cur = (rem << 64) | a.w[i];
                qq.w[i] = (uint64_t)(cur / b.w[0]);
                rem = cur % b.w[0];
            }
            qq.n = a.n;
            qq.trim();
            q = qq;
            r = BigUint((uint64_t)rem);
            return;
        }
        int s = __builtin_clzll(b.w[b.n - 1]);
        BigUint v = b << s;
        BigUint u = a << s;
        int nn = v.n, m = a.n - b.n;
        // u needs nn+m+1 limbs
        uint64_t un[CAP + 1];
        std::memset(un, 0, sizeof(un));
        for (int i = 0; i < u.n; i++) un[i] = u.w[i];
        BigUint qq;
        for (int j = m; j >= 0; j--) {
            u128 num = ((u128)un[j + nn] << 64) | un[j + nn - 1];
            u128 qhat = num / v.w[nn - 1];
            u128 rhat = num % v.w[nn - 1];
            while ((qhat >> 64) != 0 || qhat * v.w[nn - 2] > ((rhat << 64) | un[j + nn - 2])) {
                qhat--;
                rhat += v.w[nn - 1];
                if ((rhat >> 64) != 0) break;
            }
            // multiply and subtract
            u128 borrow = 0, carry = 0;
            for (int i = 0; i < nn; i++) {
                u128 p = qhat * v.w[i] + carry;
                carry = p >> 64;
                u128 t = (u128)un[i + j] - (uint64_t)p - borrow;
                un[i + j] = (uint64_t)t;
                borrow = (t >> 64) & 1;
            }
            u128 t = (u128)un[j + nn] - carry - borrow;
            un[j + nn] = (uint64_t)t;
            bool neg = (t >> 64) & 1;
            if (neg) {
                qhat--;
                u128 c = 0;
                for (int i = 0; i < nn; i++) {
                    c += (u128)un[i + j] + v.w[i];
                    un[i + j] = (uint64_t)c;
                    c >>= 64;
                }
                un[j + nn] += (uint64_t)c;
            }
            qq.w[j] = (uint64_t)qhat;
        }
        qq.n = m + 1;
        qq.trim();
        BigUint rr;
        for (int i = 0; i < nn; i++) rr.w[i] = un[i];
        rr.n = nn;
        rr.trim();
        q = qq;
        r = rr >> s;
    }
    BigUint operator/(const BigUint& o) const {
        BigUint q, r;
        div_rem(*this, o, q, r);
        return q;
    }
    BigUint operator%(const BigUint& o) const {
        BigUint q, r;
        div_rem(*this, o, q, r);
        return r;
    }

    static BigUint gcd(BigUint a, BigUint b) {
        while (!b.is_zero()) {
            BigUint t = a % b;
            a = b;
            b = t;
        }
        return a;
    }
    static BigUint lcm(const BigUint& a, const BigUint& b) { return (a / gcd(a, b)) * b; }

    static BigUint mulmod(const BigUint& a, const BigUint& b, const BigUint& m) { return (a * b) % m; }
    static BigUint powmod(const BigUint& a, const BigUint& e, const BigUint& m) {
        BigUint r(1), base = a % m;
        uint64_t nb = e.bits();
        for (uint64_t i = 0; i < nb; i++) {
            if (e.bit(i)) r = mulmod(r, base, m);
            base = mulmod(base, base, m);
        }
        return r;
    }
    // modular inverse for prime modulus; returns false when a == 0 mod p (Field::invert -> None)
    static bool invmod(const BigUint& a, const BigUint& p, BigUint& out) {
        BigUint x = a % p;
        if (x.is_zero()) return false;
        // extended Euclid with non-negative bookkeeping: track coefficients mod p
        BigUint r0 = p, r1 = x;
        BigUint t0(0), t1(1);  // t_i such that t_i * x == r_i (mod p)
        while (!r1.is_zero()) {
            BigUint q, r2;
            div_rem(r0, r1, q, r2);
            // t2 = t0 - q*t1 mod p
            BigUint qt = mulmod(q, t1, p);
            BigUint t2 = (t0 >= qt) ? (t0 - qt) : (t0 + p - qt);
            r0 = r1;
            r1 = r2;
            t0 = t1;
            t1 = t2;
        }
        assert(r0 == BigUint(1));
        out = t0;
        return true;
    }
};

}  // namespace h2o
