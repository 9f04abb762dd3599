/*
 * oracle/gl_circuit.h -- TEST INFRASTRUCTURE, not product code.
 * Flat description of the parts of plonky2's CommonCircuitData / ProverOnlyCircuitData /
 * VerifierOnlyCircuitData that prove() and verify() read (plonk/circuit_data.rs), plus the gate
 * table (the files under gates/).  The reference builds these through `builder.build::<C>()`
 * [REF src/ecdsa/gadgets/ecdsa.rs:298]; here they come from tests/synth_circuit.py.
 */
#ifndef GL_CIRCUIT_H
#define GL_CIRCUIT_H
#include "gl_field.h"

enum {
    GLO_GATE_NOOP = 0,          /* gates/noop.rs */
    GLO_GATE_CONSTANT = 1,      /* gates/constant.rs      p0 = num_consts */
    GLO_GATE_PUBLIC_INPUT = 2,  /* gates/public_input.rs */
    GLO_GATE_ARITHMETIC = 3,    /* gates/arithmetic_base.rs  p0 = num_ops */
    GLO_GATE_POSEIDON = 4,      /* gates/poseidon.rs */
    GLO_GATE_U32_INTERLEAVE = 5,      /* [REF src/u32/gates/interleave_u32.rs:33-82,230-266]   p0 = num_ops */
    GLO_GATE_UNINTERLEAVE_U32 = 6,    /* [REF src/u32/gates/uninterleave_to_u32.rs:30-91,262-309] p0 = num_ops */
    GLO_GATE_UNINTERLEAVE_B32 = 7,    /* [REF src/u32/gates/uninterleave_to_b32.rs] p0 = num_ops */
    /* gates of the secp256k1 circuit [REF src/ecdsa/gadgets/ecdsa.rs:72-96 lists them for its serializer];
     * their sources (plonky2, plonky2_u32 @552acaec) are absent: restated from the published crates */
    GLO_GATE_U32_ARITHMETIC = 8,      /* plonky2_u32 gates/arithmetic_u32.rs   p0 = num_ops */
    GLO_GATE_U32_ADD_MANY = 9,        /* plonky2_u32 gates/add_many_u32.rs     p0 = num_addends, p1 = num_ops */
    GLO_GATE_U32_SUBTRACTION = 10,    /* plonky2_u32 gates/subtraction_u32.rs  p0 = num_ops */
    GLO_GATE_U32_RANGE_CHECK = 11,    /* plonky2_u32 gates/range_check_u32.rs  p0 = num_input_limbs */
    GLO_GATE_COMPARISON = 12,         /* plonky2_u32 gates/comparison.rs       p0 = num_bits, p1 = num_chunks */
    GLO_GATE_BASE_SUM = 13,           /* plonky2 gates/base_sum.rs             p0 = num_limbs, p1 = base */
    GLO_GATE_RANDOM_ACCESS = 14,      /* plonky2 gates/random_access.rs        p0 = bits, p1 = num_copies | num_extra_constants << 16 */
};

typedef struct {
    u32 type;
    u32 selector_index;          /* which selector polynomial (column of `constants`) filters this gate */
    u32 group_start, group_end;  /* SelectorsInfo.groups[selector_index] */
    u32 row;                     /* index of the gate in CommonCircuitData.gates */
    u32 num_constraints;
    u32 p0, p1;
} glo_gate;

typedef struct {
    u32 degree_bits;
    u32 num_wires, num_routed_wires;
    u32 num_constants;           /* all constant polynomials: selectors first, then gate constants */
    u32 num_selectors;
    u32 num_challenges;
    u32 quotient_degree_factor;
    u32 num_partial_products;    /* ceil(num_routed / qdf) - 1 */
    u32 num_gate_constraints;    /* max over gates */
    u32 rate_bits, cap_height, proof_of_work_bits, num_query_rounds;
    u32 num_reductions;
    u32 reduction_arity_bits[16];
    u32 num_gates;
    u32 num_public_inputs;
    const glo_gate *gates;
    const u64 *k_is;             /* [num_routed_wires] */
    u64 circuit_digest[4];
    const u64 *constants;        /* [num_constants][n] values on H */
    const u64 *sigmas;           /* [num_routed_wires][n] values on H */
} glo_circuit;

#endif
