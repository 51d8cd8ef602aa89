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
    static void permute(uint64_t (&s)[25]) {
        static const uint64_t rc[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                                        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                                        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
        // rotation offsets r[x][y] of the rho step
        static const int rot[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
        for (int round = 0; round < 24; round++) {
            uint64_t c[5], d[5], b[25];
            for (int x = 0; x < 5; x++) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
            for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
            for (int x = 0; x < 5; x++)
                for (int y = 0; y < 5; y++) {
                    uint64_t v = s[x + 5 * y] ^ d[x];
                    b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(v, rot[x][y]);   // rho + pi
                }
            for (int y = 0; y < 5; y++)
                for (int x = 0; x < 5; x++) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);   // chi
            s[0] ^= rc[round];
        }
    }

private:
    static uint64_t rotl(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }
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
    uint8_t st_[200];
    uint8_t pos_, pos_begin_, cur_flags_;

    void run_keccak() {
        uint64_t lanes[25];
        for (int i = 0; i < 25; i++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v |= (uint64_t)st_[8 * i + j] << (8 * j); lanes[i] = v; }
        Keccak1600::permute(lanes);
        for (int i = 0; i < 25; i++) for (int j = 0; j < 8; j++) st_[8 * i + j] = (uint8_t)(lanes[i] >> (8 * j));
    }
    void run_f() {
        st_[pos_] ^= pos_begin_;
        st_[pos_ + 1] ^= 0x04;
        st_[kRate + 1] ^= 0x80;
        run_keccak();
        pos_ = 0; pos_begin_ = 0;
    }
    void absorb(const uint8_t* d, size_t n) { for (size_t i = 0; i < n; i++) { st_[pos_++] ^= d[i]; if (pos_ == kRate) run_f(); } }
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
