//! pin_fixtures.rs -- prints, from the REAL reference stack (lovesh/bulletproofs-amcl + amcl_wrapper 0.1.5 + merlin 1.x), the values
//! this repository could only restate from memory ([UNVERIFIED-RECALL]).  UNCOMPILED here: the build image has
//! no cargo / rustc.  For whoever has: copy this file to `tests/pin_fixtures.rs` of a checkout of the reference crate, then
//!
//!     cargo test --test pin_fixtures -- --nocapture --test-threads 1 | grep '^PIN ' | sed 's/^PIN //' > pins.jsonl
//!     python3 scripts/compare_pins.py pins.jsonl                      # in THIS repository
//!     (bn254:  cargo test --no-default-features --features bn254 --test pin_fixtures -- --nocapture --test-threads 1 ...)
//!
//! Every line is `PIN {"item": ..., "curve": ..., <fields>}`; byte strings are lower-case hex.  Field encodings are the ones the
//! fixtures under tests/golden/ use: field elements and points in this repository's little-endian form (BP_FMT_LE: 32-byte scalars,
//! x || y with 4 * limbs32 bytes per coordinate), PLUS the raw amcl bytes, which are the thing being pinned.
//! Items:
//!   1 fr_to_bytes       FieldElement::to_bytes(): length, byte order
//!   2 g1_to_bytes       G1::to_bytes() of the generator and of the identity (0x04 || X || Y, big-endian)
//!   3 fr_from_bytes     FieldElement::from(&[u8; MODBYTES]) of 0xff..ff (reduction mod r) and of 1
//!   4 from_msg_hash     G1::from_msg_hash for the messages of tests/golden/hash_to_g1.json; get_generators("G"/"H", 12)
//!   5 generator         G1::generator() (which BN254 this is), curve order
//!   7 transcript        the challenge bytes after committing a point and a scalar the way the crate's TranscriptProtocol does
//!                       (src/transcript.rs:47-60; the module is PRIVATE -- `mod transcript;`, src/lib.rs:23 -- so an integration test
//!                       cannot import the trait: its three one-line methods are restated below on merlin::Transcript, verbatim)
//!   8 ipp               the reference's own test_ipp instance (a = 1..4, b = 5..8, hashed generators) with y_inv FIXED to the value of
//!                       tests/golden/ipp.json "test_ipp_n4_hashed_generators": L, R, a, b and a transcript challenge afterwards
//!   9 ipp (n = 64)      BASELINE config 1's size: a = 1..64, b = 65..128, same construction, fixture "pin9_n64_hashed_generators"
#![allow(non_snake_case)]
extern crate amcl_wrapper;
extern crate bulletproofs_amcl as bp;
extern crate merlin;

use amcl_wrapper::constants::{CurveOrder, MODBYTES};
use amcl_wrapper::field_elem::{FieldElement, FieldElementVector};
use amcl_wrapper::group_elem::{GroupElement, GroupElementVector};
use amcl_wrapper::group_elem_g1::{G1Vector, G1};
use bp::ipp::IPP;
use bp::utils::get_generators;
use merlin::Transcript;

// bulletproofs_amcl::transcript is a private module (src/lib.rs:23).  What TranscriptProtocol does for these three calls, restated from
// src/transcript.rs:47-60 on the public merlin API (create_ipp itself uses the crate's own implementation internally):
fn t_commit_point(t: &mut Transcript, label: &'static [u8], p: &G1) { t.append_message(label, &p.to_bytes()); }              // :51-53
fn t_commit_scalar(t: &mut Transcript, label: &'static [u8], s: &FieldElement) { t.append_message(label, &s.to_bytes()); }  // :47-49
fn t_challenge_scalar(t: &mut Transcript, label: &'static [u8]) -> FieldElement {                                            // :55-60
    let mut buf = [0u8; MODBYTES];
    t.challenge_bytes(label, &mut buf);
    FieldElement::from(&buf)
}

#[cfg(feature = "bls381")]
const CURVE: &str = "bls12_381";
#[cfg(feature = "bn254")]
const CURVE: &str = "bn254";

fn hx(b: &[u8]) -> String { b.iter().map(|x| format!("{:02x}", x)).collect() }
fn unhx(s: &str) -> Vec<u8> { (0..s.len() / 2).map(|i| u8::from_str_radix(&s[2 * i..2 * i + 2], 16).unwrap()).collect() }

/// amcl bytes of a G1 (0x04 || X || Y big-endian) -> x || y little-endian, 4 * limbs32 bytes each (the BP_FMT_LE of this repository)
fn g1_le(p: &G1) -> String {
    if p.is_identity() { return "00".repeat(2 * MODBYTES); }
    let b = p.to_bytes();
    let (x, y) = (&b[1..1 + MODBYTES], &b[1 + MODBYTES..1 + 2 * MODBYTES]);
    let mut le: Vec<u8> = x.iter().rev().cloned().collect();
    le.extend(y.iter().rev());
    hx(&le)
}
/// 32-byte little-endian canonical scalar
fn fr_le(f: &FieldElement) -> String {
    let b = f.to_bytes();
    let le: Vec<u8> = b.iter().rev().take(32).cloned().collect();
    hx(&le)
}
fn fr_from_le(s: &str) -> FieldElement {
    let le = unhx(s);
    let mut be = vec![0u8; MODBYTES];
    for (i, v) in le.iter().enumerate() { be[MODBYTES - 1 - i] = *v; }
    FieldElement::from_bytes(&be).unwrap()
}

#[test]
fn pin_1_fr_to_bytes() {
    let one = FieldElement::from(1u64);
    let big = FieldElement::from(0x0102030405060708u64);
    println!("PIN {{\"item\": \"fr_to_bytes\", \"curve\": \"{}\", \"modbytes\": {}, \"one\": \"{}\", \"x0102030405060708\": \"{}\"}}", CURVE, MODBYTES,
             hx(&one.to_bytes()), hx(&big.to_bytes()));
}

#[test]
fn pin_2_g1_to_bytes() {
    println!("PIN {{\"item\": \"g1_to_bytes\", \"curve\": \"{}\", \"generator\": \"{}\", \"identity\": \"{}\", \"two_g\": \"{}\"}}", CURVE,
             hx(&G1::generator().to_bytes()), hx(&G1::identity().to_bytes()), hx(&G1::generator().double().to_bytes()));
}

#[test]
fn pin_3_fr_from_bytes() {
    let ff = [0xffu8; MODBYTES];
    let mut one = [0u8; MODBYTES];
    one[MODBYTES - 1] = 1;
    let mut le_one = [0u8; MODBYTES];
    le_one[0] = 1;
    println!("PIN {{\"item\": \"fr_from_bytes\", \"curve\": \"{}\", \"all_ff\": \"{}\", \"be_one\": \"{}\", \"le_one\": \"{}\"}}", CURVE,
             fr_le(&FieldElement::from(&ff)), fr_le(&FieldElement::from(&one)), fr_le(&FieldElement::from(&le_one)));
}

#[test]
fn pin_4_from_msg_hash() {
    // the messages of tests/golden/hash_to_g1.json ("from_msg_hash" cases), hex
    let msgs = ["", "67", "6730", "51", "68656c6c6f20776f726c64"];
    for m in msgs.iter() {
        let p = G1::from_msg_hash(&unhx(m));
        println!("PIN {{\"item\": \"from_msg_hash\", \"curve\": \"{}\", \"msg\": \"{}\", \"point\": \"{}\"}}", CURVE, m, g1_le(&p));
    }
    for prefix in ["G", "H"].iter() {
        let g: Vec<String> = get_generators(prefix, 12).iter().map(|p| format!("\"{}\"", g1_le(p))).collect();
        println!("PIN {{\"item\": \"get_generators\", \"curve\": \"{}\", \"prefix\": \"{}\", \"points\": [{}]}}", CURVE, prefix, g.join(", "));
    }
}

#[test]
fn pin_5_generator_and_order() {
    println!("PIN {{\"item\": \"generator\", \"curve\": \"{}\", \"G\": \"{}\", \"G_hex\": \"{}\", \"order\": \"{}\"}}", CURVE, g1_le(&G1::generator()),
             G1::generator().to_hex(), CurveOrder.tostring());
}

#[test]
fn pin_7_transcript() {
    let mut t = Transcript::new(b"pin");
    t_commit_point(&mut t, b"P", &G1::generator());
    t_commit_scalar(&mut t, b"s", &FieldElement::from(5u64));
    let c = t_challenge_scalar(&mut t, b"c");
    let mut raw = [0u8; 32];
    t.challenge_bytes(b"after", &mut raw);
    println!("PIN {{\"item\": \"transcript\", \"curve\": \"{}\", \"challenge\": \"{}\", \"after\": \"{}\"}}", CURVE, fr_le(&c), hx(&raw));
}

/// create_ipp on the reference's own test construction (src/ipp.rs:340-350) with a = first..first+n-1, b = first+n.., y_inv fixed
fn pin_ipp(name: &str, n: usize, y_inv_le: &str) {
    let a: FieldElementVector = (1..=n as u64).map(FieldElement::from).collect::<Vec<FieldElement>>().into();
    let b: FieldElementVector = ((n as u64 + 1)..=(2 * n as u64)).map(FieldElement::from).collect::<Vec<FieldElement>>().into();
    let G: G1Vector = get_generators("g", n).into();
    let H: G1Vector = get_generators("h", n).into();
    let Q = G1::from_msg_hash("Q".as_bytes());
    let G_factors: FieldElementVector = vec![FieldElement::one(); n].into();
    let y_inv = fr_from_le(y_inv_le);
    let H_factors = FieldElementVector::new_vandermonde_vector(&y_inv, n);
    let mut t = Transcript::new(b"innerproduct");
    let proof = IPP::create_ipp(&mut t, &Q, &G_factors, &H_factors, &G, &H, &a, &b);
    let mut after = [0u8; 32];
    t.challenge_bytes(b"after", &mut after);
    let l: Vec<String> = proof.L.iter().map(|p| format!("\"{}\"", g1_le(p))).collect();
    let r: Vec<String> = proof.R.iter().map(|p| format!("\"{}\"", g1_le(p))).collect();
    let l_amcl: Vec<String> = proof.L.iter().map(|p| format!("\"{}\"", hx(&p.to_bytes()))).collect();
    println!("PIN {{\"item\": \"ipp\", \"curve\": \"{}\", \"name\": \"{}\", \"L\": [{}], \"R\": [{}], \"L_amcl\": [{}], \"a_out\": \"{}\", \"b_out\": \"{}\", \"transcript_after\": \"{}\"}}",
             CURVE, name, l.join(", "), r.join(", "), l_amcl.join(", "), fr_le(&proof.a), fr_le(&proof.b), hx(&after));
}

// y_inv = H_factors[1] of the two fixtures in tests/golden/ipp.json (32-byte little-endian scalars)
#[cfg(feature = "bls381")]
const Y_INV_N4: &str = "03f67c80e40d48791d484313f1fb59636d5a80b874162c904bca2f0783c74960";
#[cfg(feature = "bls381")]
const Y_INV_N64: &str = "d12cd97125e804ff2a8fe223c436985a511819b893deec68c5dcd0a7e727411d";
#[cfg(feature = "bn254")]
const Y_INV_N4: &str = "b6f9e5e1e5c477893a62455c36db9157953364b455773fb289a2470863f1a815";
#[cfg(feature = "bn254")]
const Y_INV_N64: &str = "aa35345252185e0c2f8074cdb949bfbac1fa11473befa7ddf9b59a4323b23c15";

#[test]
fn pin_8_reference_test_ipp() { pin_ipp("test_ipp_n4_hashed_generators", 4, Y_INV_N4); }      // a = 1..4, b = 5..8: src/ipp.rs:325-389

#[test]
fn pin_9_ipp_n64() { pin_ipp("pin9_n64_hashed_generators", 64, Y_INV_N64); }                   // BASELINE config 1's size
