// bp_merlin.hpp -- host-side Fiat-Shamir transcript of the product (C++).
//
// The transcript stays on the host (L2 owns the round structure and the transcript).  The reference
// uses the crate `merlin = "1"` (Cargo.toml:10) through its TranscriptProtocol (/root/reference
// src/transcript.rs:12-61); that crate is not vendored, so this is a restatement of the published construction:
// Merlin 1.x = STROBE-128 (v1.0.2, rate 166, security 128) over Keccak-f[1600], operations meta-AD / AD / PRF.
// It reproduces the Merlin conformance vector (tests/test_host_cpu.py).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace bp {

class Keccak1600 {
public:
    // Keccak-f[1600], rounds fully unrolled on 25 named lanes (theta / rho + pi / chi / iota fused per plane).  Round 4: the first
    // version (loops with modulo indexing and a 25-lane temporary, plus a byte <-> lane repacking around every call) spent ~0.5 us per
    // permutation -- 1 ms of transcript for the 3 072 commitments of BASELINE config 3, in the prover AND in the verifier.
    static void permute(uint64_t* s) {
        static const uint64_t rc[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                                        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                                        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
        uint64_t a00 = s[0], a01 = s[1], a02 = s[2], a03 = s[3], a04 = s[4], a05 = s[5], a06 = s[6], a07 = s[7], a08 = s[8], a09 = s[9], a10 = s[10],
                 a11 = s[11], a12 = s[12], a13 = s[13], a14 = s[14], a15 = s[15], a16 = s[16], a17 = s[17], a18 = s[18], a19 = s[19], a20 = s[20],
                 a21 = s[21], a22 = s[22], a23 = s[23], a24 = s[24];
        for (int round = 0; round < 24; round++) {
            // theta
            const uint64_t c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20, c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21, c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22,
                           c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23, c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
            const uint64_t d0 = c4 ^ rotl(c1, 1), d1 = c0 ^ rotl(c2, 1), d2 = c1 ^ rotl(c3, 1), d3 = c2 ^ rotl(c4, 1), d4 = c3 ^ rotl(c0, 1);
            // rho + pi: b[y + 5 ((2x + 3y) mod 5)] = rotl(a[x + 5y] ^ d[x], r[x][y])
            const uint64_t b00 = a00 ^ d0, b10 = rotl(a01 ^ d1, 1), b20 = rotl(a02 ^ d2, 62), b05 = rotl(a03 ^ d3, 28), b15 = rotl(a04 ^ d4, 27);
            const uint64_t b16 = rotl(a05 ^ d0, 36), b01 = rotl(a06 ^ d1, 44), b11 = rotl(a07 ^ d2, 6), b21 = rotl(a08 ^ d3, 55), b06 = rotl(a09 ^ d4, 20);
            const uint64_t b07 = rotl(a10 ^ d0, 3), b17 = rotl(a11 ^ d1, 10), b02 = rotl(a12 ^ d2, 43), b12 = rotl(a13 ^ d3, 25), b22 = rotl(a14 ^ d4, 39);
            const uint64_t b23 = rotl(a15 ^ d0, 41), b08 = rotl(a16 ^ d1, 45), b18 = rotl(a17 ^ d2, 15), b03 = rotl(a18 ^ d3, 21), b13 = rotl(a19 ^ d4, 8);
            const uint64_t b14 = rotl(a20 ^ d0, 18), b24 = rotl(a21 ^ d1, 2), b09 = rotl(a22 ^ d2, 61), b19 = rotl(a23 ^ d3, 56), b04 = rotl(a24 ^ d4, 14);
            // chi (+ iota on lane 0)
            a00 = b00 ^ (~b01 & b02) ^ rc[round]; a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01);
            a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06);
            a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);
            a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);
            a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);
        }
        s[0] = a00; s[1] = a01; s[2] = a02; s[3] = a03; s[4] = a04; s[5] = a05; s[6] = a06; s[7] = a07; s[8] = a08; s[9] = a09; s[10] = a10; s[11] = a11; s[12] = a12;
        s[13] = a13; s[14] = a14; s[15] = a15; s[16] = a16; s[17] = a17; s[18] = a18; s[19] = a19; s[20] = a20; s[21] = a21; s[22] = a22; s[23] = a23; s[24] = a24;
    }

private:
    static inline uint64_t rotl(uint64_t v, int n) { return (v << n) | (v >> (64 - n)); }
};

class Strobe128 {
public:
    explicit Strobe128(const uint8_t* protocol_label, size_t len) {
        memset(st_, 0, sizeof st_);
        const uint8_t init[6] = {1, kRate + 2, 1, 0, 1, 96};
        memcpy(st_, init, 6);
        memcpy(st_ + 6, "STROBEv1.0.2", 12);
        run_keccak();
        pos_ = 0; pos_begin_ = 0; cur_flags_ = 0;
        meta_ad(protocol_label, len, false);
    }
    void meta_ad(const uint8_t* d, size_t n, bool more) { begin_op(kFlagM | kFlagA, more); absorb(d, n); }
    void ad(const uint8_t* d, size_t n, bool more) { begin_op(kFlagA, more); absorb(d, n); }
    void prf(uint8_t* out, size_t n, bool more) { begin_op(kFlagI | kFlagA | kFlagC, more); squeeze(out, n); }

private:
    static constexpr int kRate = 166;
    static constexpr uint8_t kFlagI = 1, kFlagA = 2, kFlagC = 4, kFlagT = 8, kFlagM = 16, kFlagK = 32;
    alignas(8) uint8_t st_[200];
    uint8_t pos_, pos_begin_, cur_flags_;

    void run_keccak() {
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
        uint64_t lanes[25];                            // the state bytes ARE the little-endian lanes
        memcpy(lanes, st_, 200);
        Keccak1600::permute(lanes);
        memcpy(st_, lanes, 200);
#else
        uint64_t lanes[25];
        for (int i = 0; i < 25; i++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v |= (uint64_t)st_[8 * i + j] << (8 * j); lanes[i] = v; }
        Keccak1600::permute(lanes);
        for (int i = 0; i < 25; i++) for (int j = 0; j < 8; j++) st_[8 * i + j] = (uint8_t)(lanes[i] >> (8 * j));
#endif
    }
    void run_f() {
        st_[pos_] ^= pos_begin_;
        st_[pos_ + 1] ^= 0x04;
        st_[kRate + 1] ^= 0x80;
        run_keccak();
        pos_ = 0; pos_begin_ = 0;
    }
    void absorb(const uint8_t* d, size_t n) {
        while (n) {
            size_t k = (size_t)(kRate - pos_) < n ? (size_t)(kRate - pos_) : n;
            for (size_t i = 0; i < k; i++) st_[pos_ + i] ^= d[i];
            pos_ = (uint8_t)(pos_ + k); d += k; n -= k;
            if (pos_ == kRate) run_f();
        }
    }
    void squeeze(uint8_t* d, size_t n) { for (size_t i = 0; i < n; i++) { d[i] = st_[pos_]; st_[pos_++] = 0; if (pos_ == kRate) run_f(); } }
    void begin_op(uint8_t flags, bool more) {
        if (more) return;   // continuation: same flags as the operation in progress
        uint8_t hdr[2] = {pos_begin_, flags};
        pos_begin_ = (uint8_t)(pos_ + 1);
        cur_flags_ = flags;
        absorb(hdr, 2);
        if ((flags & (kFlagC | kFlagK)) && pos_ != 0) run_f();
    }
};

// merlin::Transcript
class Transcript {
public:
    Transcript(const uint8_t* label, size_t len) : strobe_((const uint8_t*)"Merlin v1.0", 11) { append_message((const uint8_t*)"dom-sep", 7, label, len); }
    void append_message(const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen) {
        uint8_t le[4] = {(uint8_t)mlen, (uint8_t)(mlen >> 8), (uint8_t)(mlen >> 16), (uint8_t)(mlen >> 24)};
        strobe_.meta_ad(label, llen, false);
        strobe_.meta_ad(le, 4, true);
        strobe_.ad(msg, mlen, false);
    }
    void append_u64(const uint8_t* label, size_t llen, uint64_t x) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
        append_message(label, llen, b, 8);
    }
    void challenge_bytes(const uint8_t* label, size_t llen, uint8_t* out, size_t olen) {
        uint8_t le[4] = {(uint8_t)olen, (uint8_t)(olen >> 8), (uint8_t)(olen >> 16), (uint8_t)(olen >> 24)};
        strobe_.meta_ad(label, llen, false);
        strobe_.meta_ad(le, 4, true);
        strobe_.prf(out, olen, false);
    }

private:
    Strobe128 strobe_;
};

}  // namespace bp
