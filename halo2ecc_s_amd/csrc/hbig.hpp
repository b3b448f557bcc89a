// Host-side arbitrary precision unsigned integers, used only at set-up time to derive the
// field-pair constants (RangeInfo, src/range_info.rs:77-184, :334-359) and the fixed-column
// dictionary.  Not on the hot path; simplicity over speed.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

namespace h2e {

struct HBig {
    std::vector<uint64_t> w;  // little endian, no trailing zeros

    HBig() {}
    HBig(uint64_t v) {
        if (v) w.push_back(v);
    }
    static HBig from_words(const uint64_t* p, int n) {
        HBig r;
        r.w.assign(p, p + n);
        r.trim();
        return r;
    }
    static HBig from_hex(const std::string& s) {
        HBig r;
        for (char c : s) {
            int v = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : c - 'A' + 10;
            r = r.shl(4) + HBig((uint64_t)v);
        }
        return r;
    }
    void trim() {
        while (!w.empty() && w.back() == 0) w.pop_back();
    }
    bool is_zero() const { return w.empty(); }
    int bits() const {
        if (w.empty()) return 0;
        return (int)(w.size() - 1) * 64 + (64 - __builtin_clzll(w.back()));
    }
    bool bit(int i) const {
        size_t k = (size_t)i / 64;
        return k < w.size() && ((w[k] >> (i % 64)) & 1);
    }
    uint64_t word(size_t i) const { return i < w.size() ? w[i] : 0; }
    void to_words(uint64_t* out, int n) const {
        for (int i = 0; i < n; i++) out[i] = word(i);
        if ((int)w.size() > n) throw std::runtime_error("HBig::to_words overflow");
    }
    static int cmp(const HBig& a, const HBig& b) {
        if (a.w.size() != b.w.size()) return a.w.size() < b.w.size() ? -1 : 1;
        for (size_t i = a.w.size(); i-- > 0;)
            if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
        return 0;
    }
    bool operator<(const HBig& o) const { return cmp(*this, o) < 0; }
    bool operator<=(const HBig& o) const { return cmp(*this, o) <= 0; }
    bool operator>=(const HBig& o) const { return cmp(*this, o) >= 0; }
    bool operator>(const HBig& o) const { return cmp(*this, o) > 0; }
    bool operator==(const HBig& o) const { return cmp(*this, o) == 0; }
    HBig operator+(const HBig& o) const {
        HBig r;
        size_t n = std::max(w.size(), o.w.size());
        unsigned __int128 c = 0;
        for (size_t i = 0; i < n; i++) {
            c += (unsigned __int128)word(i) + o.word(i);
            r.w.push_back((uint64_t)c);
            c >>= 64;
        }
        if (c) r.w.push_back((uint64_t)c);
        return r;
    }
    HBig operator-(const HBig& o) const {
        if (cmp(*this, o) < 0) throw std::runtime_error("HBig underflow");
        HBig r;
        uint64_t borrow = 0;
        for (size_t i = 0; i < w.size(); i++) {
            unsigned __int128 t = (unsigned __int128)w[i] - o.word(i) - borrow;
            r.w.push_back((uint64_t)t);
            borrow = (uint64_t)(t >> 64) & 1;
        }
        r.trim();
        return r;
    }
    HBig operator*(const HBig& o) const {
        HBig r;
        if (is_zero() || o.is_zero()) return r;
        r.w.assign(w.size() + o.w.size(), 0);
        for (size_t i = 0; i < w.size(); i++) {
            unsigned __int128 c = 0;
            for (size_t j = 0; j < o.w.size(); j++) {
                c += (unsigned __int128)w[i] * o.w[j] + r.w[i + j];
                r.w[i + j] = (uint64_t)c;
                c >>= 64;
            }
            r.w[i + o.w.size()] = (uint64_t)c;
        }
        r.trim();
        return r;
    }
    HBig shl(int s) const {
        HBig r;
        if (is_zero()) return r;
        int ws = s / 64, bs = s % 64;
        r.w.assign(w.size() + ws + 1, 0);
        for (size_t i = 0; i < w.size(); i++) {
            r.w[i + ws] |= w[i] << bs;
            if (bs) r.w[i + ws + 1] |= w[i] >> (64 - bs);
        }
        r.trim();
        return r;
    }
    HBig shr(int s) const {
        HBig r;
        int ws = s / 64, bs = s % 64;
        if ((size_t)ws >= w.size()) return r;
        r.w.assign(w.size() - ws, 0);
        for (size_t i = ws; i < w.size(); i++) {
            r.w[i - ws] = w[i] >> bs;
            if (bs && i + 1 < w.size()) r.w[i - ws] |= w[i + 1] << (64 - bs);
        }
        r.trim();
        return r;
    }
    HBig low_bits(int n) const {  // self mod 2^n
        HBig r = *this;
        size_t words = (size_t)(n + 63) / 64;
        if (r.w.size() > words) r.w.resize(words);
        if (n % 64 && r.w.size() == words) r.w[words - 1] &= (1ull << (n % 64)) - 1;
        r.trim();
        return r;
    }
    // shift-subtract long division
    static void divmod(const HBig& a, const HBig& b, HBig& q, HBig& r) {
        if (b.is_zero()) throw std::runtime_error("HBig div by zero");
        q = HBig();
        r = HBig();
        for (int i = a.bits() - 1; i >= 0; i--) {
            r = r.shl(1);
            if (a.bit(i)) r = r + HBig(1);
            if (r >= b) {
                r = r - b;
                size_t k = (size_t)i / 64;
                if (q.w.size() <= k) q.w.resize(k + 1, 0);
                q.w[k] |= 1ull << (i % 64);
            }
        }
        q.trim();
    }
    HBig operator/(const HBig& o) const {
        HBig q, r;
        divmod(*this, o, q, r);
        return q;
    }
    HBig operator%(const HBig& o) const {
        HBig q, r;
        divmod(*this, o, q, r);
        return r;
    }
};

}  // namespace h2e
