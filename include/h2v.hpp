// C++ host-side mirror of the reference's surface for the accelerated path, over the C ABI in h2v.h.
//
// The reference is a Rust crate; a Rust toolchain is not available where this library is built, so the host side
// above the C ABI is offered in C++ (this header; header-only, C++17) and in Python (halo2_verifier_amd/verifier.py),
// and as uncompiled Rust source in integration/rust/.  Names, argument meaning and error behaviour follow the reference:
//
//   reference (halo2_verifier)                                   here (namespace halo2_verifier)
//   ---------------------------------------------------------    ---------------------------------------------
//   helpers::SerdeFormat                  helpers.rs:7-19         SerdeFormat
//   plonk::Error                          plonk/mod.rs:19-32      Error (same order: InvalidInstances = -1 ...)
//   ParamsKZG::read_custom                kzg/commitment.rs:155   ParamsKZG(bytes, format)
//   VerifyingKey::read                    plonk/vk.rs:76-115      VerifyingKey(bytes, format)
//   verify_proof(params, vk, strategy, instances, transcript)     verify_proof(params, vk, strategy, instances, proof)
//                                         lib.rs:33-49
//   AccumulatorStrategy::{new, process, finalize}                 AccumulatorStrategy(params): verify_proof() queues,
//                                         kzg/strategy.rs:99-141    finalize() runs the batch on the GPU
//   SingleStrategy                        kzg/strategy.rs:143-181 SingleStrategy(params): verify_proof() runs at once
//   VerifierSHPLONK / VerifierGWC, Blake2bRead / Keccak256Read    MultiOpen, TranscriptKind (generic parameters of lib.rs:33-40)
//
// There is no CPU fallback: constructing a Context without a HIP device throws Failure{H2V_ERR_DEVICE}.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "h2v.h"

namespace halo2_verifier {

enum class SerdeFormat : int { Processed = H2V_SERDE_PROCESSED, RawBytes = H2V_SERDE_RAW_BYTES, RawBytesUnchecked = H2V_SERDE_RAW_BYTES_UNCHECKED };
enum class MultiOpen : int { SHPLONK = H2V_MULTIOPEN_SHPLONK, GWC = H2V_MULTIOPEN_GWC };
enum class TranscriptKind : int { Blake2b = H2V_TRANSCRIPT_BLAKE2B, Keccak256 = H2V_TRANSCRIPT_KECCAK256 };

// plonk::Error (plonk/mod.rs:19-32), in declaration order
enum class Error : int {
    Ok = 0,
    InvalidInstances = H2V_ERR_INVALID_INSTANCES,
    ConstraintSystemFailure = H2V_ERR_CONSTRAINT_SYSTEM_FAILURE,
    BoundsFailure = H2V_ERR_BOUNDS_FAILURE,
    Opening = H2V_ERR_OPENING,
    Transcript = H2V_ERR_TRANSCRIPT,
    InstanceTooLarge = H2V_ERR_INSTANCE_TOO_LARGE,
};

// a failure of the library itself (bad argument, malformed VK / params, no device), as opposed to a proof that does not verify
struct Failure : std::runtime_error {
    int code;
    Failure(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc) { if (rc != 0) throw Failure(rc, h2v_last_error()); }

typedef std::vector<uint8_t> Bytes;
typedef std::vector<Bytes> Column;            // one instance column: 32-byte little-endian canonical Fr values
typedef std::vector<Column> Instances;        // one circuit instance: its columns (the reference's &[&[Fr]]); with circuit_instances = M
                                              // (lib.rs:43: instances.len()) the M x columns of the transcript's instances, instance by instance

struct ParamsKZG { Bytes bytes; SerdeFormat format = SerdeFormat::RawBytes; };
struct VerifyingKey { Bytes bytes; SerdeFormat format = SerdeFormat::RawBytes; };

// ParamsKZG + VerifyingKey resident on one GPU (h2v_ctx)
class Context {
public:
    Context(const ParamsKZG& p, const VerifyingKey& vk, int device = 0, MultiOpen mo = MultiOpen::SHPLONK, TranscriptKind tr = TranscriptKind::Blake2b,
            int circuit_instances = 1) {
        h2v_options o = H2V_OPTIONS_INIT;
        o.multiopen = (int)mo; o.transcript = (int)tr; o.circuit_instances = circuit_instances;
        check(h2v_ctx_create_ex(p.bytes.data(), p.bytes.size(), (int)p.format, vk.bytes.data(), vk.bytes.size(), (int)vk.format, device, &o, &h_));
    }
    ~Context() { if (h_) h2v_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    h2v_ctx* handle() const { return h_; }

private:
    h2v_ctx* h_ = nullptr;
};

namespace detail {
// pointer-array view of (proof, instances) pairs in the layout h2v_verify_batch / h2v_verify_each take
struct Packed {
    std::vector<const uint8_t*> proofs, insts;
    std::vector<size_t> lens, col_lens, col_lens_per_proof;   // col_lens: proof 0's shape; per proof: [n][cols]
    bool uniform = true;                                        // every proof has proof 0's instance shape
    std::vector<Bytes> flat;
    Packed(const std::vector<std::pair<Instances, Bytes>>& items, size_t ncols_if_empty) {
        for (const auto& it : items) {
            Bytes f;
            for (const Column& c : it.first) for (const Bytes& v : c) f.insert(f.end(), v.begin(), v.end());
            flat.push_back(std::move(f));
        }
        for (size_t i = 0; i < items.size(); ++i) {
            proofs.push_back(items[i].second.data()); lens.push_back(items[i].second.size()); insts.push_back(flat[i].data());
        }
        if (!items.empty()) for (const Column& c : items[0].first) col_lens.push_back(c.size());
        else col_lens.assign(ncols_if_empty, 0);
        for (const auto& it : items) {
            if (it.first.size() != col_lens.size()) throw Failure(H2V_ERR_INVALID_INSTANCES, "instances do not match the VK's instance column count");
            for (size_t c = 0; c < col_lens.size(); ++c) {
                if (it.first[c].size() != col_lens[c]) uniform = false;      // verify_proof takes `instances` per call: shapes may differ
                col_lens_per_proof.push_back(it.first[c].size());
            }
        }
    }
};
}  // namespace detail

// kzg/strategy.rs:99-141: verify_proof() adds a proof to the accumulator, finalize() runs ONE pairing for all of them.
// Here the proofs are queued on the host and the whole batch runs on the GPU at finalize().
class AccumulatorStrategy {
public:
    explicit AccumulatorStrategy(const ParamsKZG& p, int device = 0, MultiOpen mo = MultiOpen::SHPLONK, TranscriptKind tr = TranscriptKind::Blake2b,
                                 int circuit_instances = 1)
        : params_(p), device_(device), mo_(mo), tr_(tr), ci_(circuit_instances) {}
    // AccumulatorStrategy::with(msm_accumulator) (kzg/strategy.rs:75-78): start from an existing DualMSM; the channels are term lists
    // as MSMKZG holds them — scalars 32 bytes each, bases 64 bytes (x | y) each.  A finished accumulation resumes with scalar 1 and
    // its evaluated channels (left(), right()) as the single base of either side.
    static AccumulatorStrategy with(const ParamsKZG& p, Bytes left_scalars, Bytes left_bases, Bytes right_scalars, Bytes right_bases, int device = 0,
                                    MultiOpen mo = MultiOpen::SHPLONK, TranscriptKind tr = TranscriptKind::Blake2b, int circuit_instances = 1) {
        if (left_scalars.size() % 32 || left_bases.size() != 2 * left_scalars.size() || right_scalars.size() % 32 || right_bases.size() != 2 * right_scalars.size())
            throw Failure(H2V_ERR_BAD_ARGUMENT, "a seed channel is n 32-byte scalars and n 64-byte bases");
        AccumulatorStrategy s(p, device, mo, tr, circuit_instances);
        s.seeded_ = true;
        s.seed_ls_ = std::move(left_scalars); s.seed_lb_ = std::move(left_bases); s.seed_rs_ = std::move(right_scalars); s.seed_rb_ = std::move(right_bases);
        return s;
    }
    // rand32: the Fr::random draws of process() (kzg/strategy.rs:129), one 32-byte canonical scalar per proof; empty = OS RNG
    void set_randomness(Bytes rand32) { rand_ = std::move(rand32); }
    void push(const VerifyingKey& vk, Instances inst, Bytes proof) {
        if (!items_.empty() && vk.bytes != vk_.bytes) throw Failure(H2V_ERR_BAD_ARGUMENT, "one AccumulatorStrategy batch verifies proofs of one VerifyingKey");
        vk_ = vk;
        items_.emplace_back(std::move(inst), std::move(proof));
    }
    // -> true iff every verify_proof succeeded and the pairing check passed; statuses() then holds the per-proof plonk::Error
    bool finalize() {
        Context ctx(params_, vk_, device_, mo_, tr_, ci_);
        size_t ncols = 0;
        check(h2v_ctx_proof_shape(ctx.handle(), nullptr, nullptr, nullptr, nullptr, &ncols));
        detail::Packed pk(items_, ncols);
        statuses_.assign(items_.size() ? items_.size() : 1, 0);
        int ok = 0;
        if (!rand_.empty() && rand_.size() != 32 * items_.size()) throw Failure(H2V_ERR_BAD_ARGUMENT, "one 32-byte draw per proof");
        if (seeded_) {
            if (!pk.uniform) throw Failure(H2V_ERR_UNSUPPORTED, "a seeded accumulation takes one instance shape");
            check(h2v_verify_batch_seeded(ctx.handle(), items_.size(), pk.proofs.data(), pk.lens.data(), pk.insts.data(), pk.col_lens.size(), pk.col_lens.data(),
                                          rand_.empty() ? nullptr : rand_.data(), seed_ls_.data(), seed_lb_.data(), seed_ls_.size() / 32, seed_rs_.data(), seed_rb_.data(),
                                          seed_rs_.size() / 32, statuses_.data(), &ok, left_, right_));
        } else if (pk.uniform)
            check(h2v_verify_batch(ctx.handle(), items_.size(), pk.proofs.data(), pk.lens.data(), pk.insts.data(), pk.col_lens.size(), pk.col_lens.data(),
                                   rand_.empty() ? nullptr : rand_.data(), statuses_.data(), &ok, left_, right_));
        else
            check(h2v_verify_batch_shapes(ctx.handle(), items_.size(), pk.proofs.data(), pk.lens.data(), pk.insts.data(), pk.col_lens.size(), pk.col_lens_per_proof.data(),
                                          rand_.empty() ? nullptr : rand_.data(), statuses_.data(), &ok, left_, right_));
        statuses_.resize(items_.size());
        return ok != 0;
    }
    const std::vector<int>& statuses() const { return statuses_; }
    const uint8_t* left() const { return left_; }     // evaluated channels of the final DualMSM, canonical x|y
    const uint8_t* right() const { return right_; }

private:
    ParamsKZG params_; VerifyingKey vk_; int device_; MultiOpen mo_; TranscriptKind tr_; int ci_;
    std::vector<std::pair<Instances, Bytes>> items_;
    Bytes rand_;
    bool seeded_ = false;
    Bytes seed_ls_, seed_lb_, seed_rs_, seed_rb_;
    std::vector<int> statuses_;
    uint8_t left_[64] = {0}, right_[64] = {0};
};

// kzg/strategy.rs:143-181: one pairing per proof, checked inside verify_proof
class SingleStrategy {
public:
    explicit SingleStrategy(const ParamsKZG& p, int device = 0, MultiOpen mo = MultiOpen::SHPLONK, TranscriptKind tr = TranscriptKind::Blake2b,
                            int circuit_instances = 1)
        : params_(p), device_(device), mo_(mo), tr_(tr), ci_(circuit_instances) {}
    Error verify(const VerifyingKey& vk, const Instances& inst, const Bytes& proof) const {
        Context ctx(params_, vk, device_, mo_, tr_, ci_);
        std::vector<std::pair<Instances, Bytes>> one{{inst, proof}};
        size_t ncols = 0;
        check(h2v_ctx_proof_shape(ctx.handle(), nullptr, nullptr, nullptr, nullptr, &ncols));
        if (inst.size() != ncols) return Error::InvalidInstances;   // lib.rs:51-55
        detail::Packed pk(one, ncols);
        int st = 0;
        check(h2v_verify_each(ctx.handle(), 1, pk.proofs.data(), pk.lens.data(), pk.insts.data(), pk.col_lens.size(), pk.col_lens.data(), &st));
        return (Error)st;
    }

private:
    ParamsKZG params_; int device_; MultiOpen mo_; TranscriptKind tr_; int ci_;
};

// lib.rs:33-49.  SingleStrategy: returns the proof's plonk::Error (Ok = accepted).
inline Error verify_proof(const ParamsKZG&, const VerifyingKey& vk, const SingleStrategy& s, const Instances& inst, const Bytes& proof) { return s.verify(vk, inst, proof); }
// AccumulatorStrategy: Output = the strategy (the proof is queued; errors surface in statuses() after finalize())
inline AccumulatorStrategy& verify_proof(const ParamsKZG&, const VerifyingKey& vk, AccumulatorStrategy& s, Instances inst, Bytes proof) {
    s.push(vk, std::move(inst), std::move(proof));
    return s;
}

}  // namespace halo2_verifier
