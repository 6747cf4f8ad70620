// C++ caller of the library through include/h2v.hpp — the compiled-language counterpart of the reference's own tests
// (halo2_verifier/tests/helpers.rs:66-85: verify_proof with SingleStrategy; an AccumulatorStrategy batch; tampered inputs).
//
//   harness <dir>
// reads <dir>/params.bin, vk.bin, proofs.bin (n x proof_len), inst.bin (n x n_pub x 32), rand.bin (n x 32), meta.txt
// ("n proof_len n_pub"), runs the same calls a Rust user of the crate would make, and prints one line per result:
//   single <i> <plonk::Error as int>
//   batch <ok 0/1> <left hex> <right hex> <status...>
// tests/test_gpu_cpp_harness.py builds it with g++, feeds it seeded inputs and compares every line with the CPU oracle.
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include "../../include/h2v.hpp"

using namespace halo2_verifier;

static Bytes slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot read %s\n", p.c_str()); exit(2); }
    return Bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void hex(const uint8_t* b, size_t n) { for (size_t i = 0; i < n; ++i) printf("%02x", b[i]); }

int main(int argc, char** argv) {
    if (argc != 2) { fprintf(stderr, "usage: harness <dir>\n"); return 2; }
    const std::string d = argv[1];
    size_t n = 0, proof_len = 0, n_pub = 0;
    { FILE* m = fopen((d + "/meta.txt").c_str(), "r"); if (!m || fscanf(m, "%zu %zu %zu", &n, &proof_len, &n_pub) != 3) return 2; fclose(m); }
    try {
        ParamsKZG params{slurp(d + "/params.bin"), SerdeFormat::RawBytes};
        VerifyingKey vk{slurp(d + "/vk.bin"), SerdeFormat::RawBytes};
        Bytes proofs = slurp(d + "/proofs.bin"), inst = slurp(d + "/inst.bin"), rand = slurp(d + "/rand.bin");
        auto proof_of = [&](size_t i) { return Bytes(proofs.begin() + i * proof_len, proofs.begin() + (i + 1) * proof_len); };
        auto inst_of = [&](size_t i) {
            Column c;
            for (size_t j = 0; j < n_pub; ++j) c.emplace_back(inst.begin() + (i * n_pub + j) * 32, inst.begin() + (i * n_pub + j + 1) * 32);
            return Instances{c};
        };
        // let strategy = SingleStrategy::new(&params); verify_proof(&params, &vk, strategy, &[&[&pubs]], &mut transcript)
        SingleStrategy single(params);
        for (size_t i = 0; i < n; ++i) printf("single %zu %d\n", i, (int)verify_proof(params, vk, single, inst_of(i), proof_of(i)));
        // wrong number of instance columns: Error::InvalidInstances (lib.rs:51-55)
        printf("single_bad_columns %d\n", (int)verify_proof(params, vk, single, Instances{}, proof_of(0)));
        // let mut s = AccumulatorStrategy::new(&params); for each proof { s = verify_proof(.., s, ..)? } s.finalize()
        AccumulatorStrategy acc(params);
        acc.set_randomness(rand);
        for (size_t i = 0; i < n; ++i) verify_proof(params, vk, acc, inst_of(i), proof_of(i));
        const bool ok = acc.finalize();
        printf("batch %d ", ok ? 1 : 0); hex(acc.left(), 64); printf(" "); hex(acc.right(), 64);
        for (int s : acc.statuses()) printf(" %d", s);
        printf("\n");
        // AccumulatorStrategy::with(msm_accumulator) (kzg/strategy.rs:75-78): the first half accumulated, its evaluated channels
        // resumed as the seed of the second half — must equal the one accumulation above
        const size_t half = n / 2;
        AccumulatorStrategy first(params);
        first.set_randomness(Bytes(rand.begin(), rand.begin() + 32 * half));
        for (size_t i = 0; i < half; ++i) verify_proof(params, vk, first, inst_of(i), proof_of(i));
        const bool ok1 = first.finalize();
        Bytes one(32, 0); one[0] = 1;
        AccumulatorStrategy second = AccumulatorStrategy::with(params, one, Bytes(first.left(), first.left() + 64), one, Bytes(first.right(), first.right() + 64));
        second.set_randomness(Bytes(rand.begin() + 32 * half, rand.end()));
        for (size_t i = half; i < n; ++i) verify_proof(params, vk, second, inst_of(i), proof_of(i));
        const bool ok2 = second.finalize();
        printf("resumed %d ", (ok1 && ok2) ? 1 : 0); hex(second.left(), 64); printf(" "); hex(second.right(), 64);
        printf("\n");
    } catch (const Failure& e) {
        printf("failure %d %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
