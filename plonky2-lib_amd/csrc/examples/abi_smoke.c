/* abi_smoke.c -- a plain C99 consumer of include/glp.h: what a cgo / Rust-FFI / JNI binding sees.
 * Builds against libglprover.so with no C++ and no HIP headers; on a machine with an MI355X it commits a small
 * PolynomialBatch and prints the Poseidon known-answer vector the reference checks
 * [REF src/zkdsa/circuits/mod.rs:85-101]; without a GPU it reports the library's error and exits 2.
 *   gcc -std=c99 -Wall -Werror -I../../../include abi_smoke.c -L../.. -lglprover -Wl,-rpath,'$ORIGIN/../..' -o abi_smoke */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "glp.h"

/* argv[1] (optional): a circuit hand-off file (glp_circuit_file_*, host-only code: this part runs without a GPU).  With a GPU the
 * circuit and witness of the file are proved through glp_circuit_create / glp_prove / glp_verify. */
int main(int argc, char **argv) {
    glp_circuit_file *cf = NULL;
    if (argc > 1) {
        if (glp_circuit_file_open(argv[1], 1, &cf) != GLP_OK) { fprintf(stderr, "glp_circuit_file_open: %s\n", glp_last_error()); return 1; }
        const glp_circuit_desc *d = glp_circuit_file_desc(cf);
        printf("circuit file ok: 2^%u rows, %u wires, %u gates, witness %s\n", d->degree_bits, d->num_wires, d->num_gates,
               glp_circuit_file_wires(cf) ? "present" : "absent");
    }
    glp_ctx *ctx = NULL;
    if (glp_ctx_create(0, &ctx) != GLP_OK) {
        fprintf(stderr, "glp_ctx_create: %s\n", glp_last_error());
        return 2;
    }
    uint64_t state[12] = {0};
    if (glp_poseidon_permute(ctx, state, 1) != GLP_OK) { fprintf(stderr, "%s\n", glp_last_error()); return 1; }
    printf("two_to_one(0,0) = [%llu, %llu, %llu, %llu]\n", (unsigned long long)state[0], (unsigned long long)state[1],
           (unsigned long long)state[2], (unsigned long long)state[3]);
    const uint64_t kat[4] = {4330397376401421145ull, 14124799381142128323ull, 8742572140681234676ull, 14345658006221440202ull};
    for (int i = 0; i < 4; i++)
        if (state[i] != kat[i]) { fprintf(stderr, "Poseidon known-answer mismatch\n"); return 1; }

    enum { NCOLS = 4, LOG_N = 8, CAP_H = 2 };
    uint64_t *vals = (uint64_t *)malloc(sizeof(uint64_t) * NCOLS << LOG_N);
    for (size_t i = 0; i < (size_t)NCOLS << LOG_N; i++) vals[i] = (uint64_t)i * 0x9E3779B97F4A7C15ull % 0xFFFFFFFF00000001ull;
    glp_batch *b = NULL;
    if (glp_batch_from_values(ctx, vals, NCOLS, LOG_N, 3, CAP_H, &b) != GLP_OK) { fprintf(stderr, "%s\n", glp_last_error()); return 1; }
    uint64_t cap[(1 << CAP_H) * 4];
    if (glp_batch_cap(b, cap) != GLP_OK) { fprintf(stderr, "%s\n", glp_last_error()); return 1; }
    printf("cap[0] = [%llu, %llu, %llu, %llu]\n", (unsigned long long)cap[0], (unsigned long long)cap[1], (unsigned long long)cap[2],
           (unsigned long long)cap[3]);
    glp_batch_free(b);
    free(vals);
    if (cf && glp_circuit_file_wires(cf)) {
        glp_circuit *circuit = NULL;
        if (glp_circuit_create(ctx, glp_circuit_file_desc(cf), &circuit) != GLP_OK) { fprintf(stderr, "%s\n", glp_last_error()); return 1; }
        const size_t words = glp_proof_words(circuit);
        uint64_t *proof = (uint64_t *)malloc(sizeof(uint64_t) * words);
        if (glp_prove(ctx, circuit, glp_circuit_file_wires(cf), glp_circuit_file_public_inputs(cf), proof) != GLP_OK ||
            glp_verify_n(circuit, proof, words) != GLP_OK) { fprintf(stderr, "%s\n", glp_last_error()); return 1; }
        printf("proved and verified the circuit of %s (%zu proof words)\n", argv[1], words);
        /* the pipelined hand-over: the witness in page-locked memory, staged on the copy stream, proved from the staged copy; then both
         * proofs and a damaged one through the batch verifier (query rounds on the GPU) */
        const glp_circuit_desc *d = glp_circuit_file_desc(cf);
        const size_t wbytes = sizeof(uint64_t) * d->num_wires << d->degree_bits;
        void *pinned = NULL;
        glp_witness *staged = NULL;
        uint64_t *three = (uint64_t *)malloc(sizeof(uint64_t) * words * 3);
        if (glp_host_alloc(ctx, wbytes, &pinned) != GLP_OK) { fprintf(stderr, "%s\n", glp_last_error()); return 1; }
        memcpy(pinned, glp_circuit_file_wires(cf), wbytes);
        if (glp_witness_stage(ctx, circuit, (const uint64_t *)pinned, 0, &staged) != GLP_OK ||
            glp_prove_staged(ctx, circuit, staged, glp_circuit_file_public_inputs(cf), three + words) != GLP_OK) {
            fprintf(stderr, "%s\n", glp_last_error());
            return 1;
        }
        glp_witness_free(staged);
        glp_host_free(ctx, pinned);
        if (memcmp(proof, three + words, sizeof(uint64_t) * words) != 0) { fprintf(stderr, "the staged proof differs from glp_prove's\n"); return 1; }
        memcpy(three, proof, sizeof(uint64_t) * words);
        memcpy(three + 2 * words, proof, sizeof(uint64_t) * words);
        three[2 * words + words / 2] ^= 1;
        int32_t status[3];
        char reasons[3 * GLP_REASON_LEN];
        if (glp_verify_batch(ctx, circuit, 3, three, status, reasons) != GLP_OK) { fprintf(stderr, "%s\n", glp_last_error()); return 1; }
        if (status[0] != GLP_OK || status[1] != GLP_OK || status[2] != GLP_ERR_PROVE || reasons[0] || !reasons[2 * GLP_REASON_LEN]) {
            fprintf(stderr, "glp_verify_batch verdicts %d %d %d\n", status[0], status[1], status[2]);
            return 1;
        }
        printf("staged proof equal; batch verifier: ok, ok, rejected (%s)\n", reasons + 2 * GLP_REASON_LEN);
        free(three);
        free(proof);
        glp_circuit_free(circuit);
    }
    glp_circuit_file_close(cf);
    glp_ctx_destroy(ctx);
    printf("abi_smoke ok (%s)\n", glp_version());
    return 0;
}
