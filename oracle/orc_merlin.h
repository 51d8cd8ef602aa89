/* oracle/orc_merlin.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * Merlin 1.x transcript = STROBE-128 (v1.0.2) over Keccak-f[1600], restated from the published
 * specifications (merlin.cool, strobe.sourceforge.io): the `merlin = "1"` crate the reference
 * depends on (Cargo.toml:10) is not vendored.  Pinned by the Merlin conformance vector
 * ("test protocol" / "some label" / "some data" -> d5a21972...0615) in tests/golden/merlin.json.
 * Call sites restated: src/transcript.rs:29-61.
 */
#ifndef ORC_MERLIN_H
#define ORC_MERLIN_H
#include <stddef.h>
#include <stdint.h>

typedef struct {
    uint8_t st[200];
    uint8_t pos, pos_begin, cur_flags;
} orc_transcript;

void orc_keccak_f1600(uint8_t st[200]);
/* FIPS 202 SHAKE256 (rate 136, suffix 0x1f) -- amcl_wrapper `hash_msg` uses it for G1::from_msg_hash */
void orc_shake256(const uint8_t* msg, size_t len, uint8_t* out, size_t out_len);
void orc_transcript_init(orc_transcript* t, const uint8_t* label, size_t label_len);
void orc_transcript_append(orc_transcript* t, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len);
void orc_transcript_append_u64(orc_transcript* t, const uint8_t* label, size_t label_len, uint64_t x);
void orc_transcript_challenge(orc_transcript* t, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len);

#endif
