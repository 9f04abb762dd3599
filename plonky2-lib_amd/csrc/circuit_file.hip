// circuit_file.hip -- the circuit hand-off file: what a machine WITH the Rust builder writes and
// the GPU box reads.  Host code only (no device work): a flat, versioned dump of glp_circuit_desc -- the parts of
// plonky2's CommonCircuitData / ProverOnlyCircuitData that prove() reads, i.e. what `builder.build::<C>()` returns at
// [REF src/ecdsa/gadgets/ecdsa.rs:298] and what the reference itself round-trips through `CircuitData::to_bytes` /
// `from_bytes` at [REF src/ecdsa/gadgets/ecdsa.rs:298-316] with its gate / generator serializer tables
// [REF src/ecdsa/gadgets/ecdsa.rs:68-135, src/ecdsa/serialization.rs:7-46] -- optionally followed by one witness
// (`PartitionWitness` wire values after generate_partial_witness) and its public inputs.
//
// This is NOT plonky2's own `Buffer` layout (that needs the generator serializers and is recalled, not pinned): it is
// the documented layout below, which a ~40-line Rust writer fills from `data.common` / `data.prover_only`
// (INTEGRATION.md).  Its purpose: let a Rust machine produce the (circuit, witness, proof) fixture that finally pins
// proof bytes, and let real circuits reach the GPU prover without an in-process descriptor.
//
// Layout (all integers little-endian; every section starts on an 8-byte boundary):
//   0   char[8]  magic "GLPCIRC1"
//   8   u32      version = 2 (version 1 files -- checksum over the sections only -- are still read)
//   12  u32      header_bytes (offset of the first section)
//   16  u32[14]  degree_bits, num_wires, num_routed_wires, num_constants, num_selectors, num_challenges,
//                quotient_degree_factor, num_partial_products, num_gate_constraints, rate_bits, cap_height,
//                proof_of_work_bits, num_query_rounds, num_reductions
//   72  u32[16]  reduction_arity_bits
//   136 u32      num_gates
//   140 u32      num_public_inputs
//   144 u32      has_witness (0 / 1)
//   148 u32      hasher (GLP_HASH_POSEIDON = 0, GLP_HASH_KECCAK25 = 1; files written before the field existed hold 0)
//   152 u64[4]   circuit_digest (all zero = derive it from the constants/sigmas cap)
//   184 u64      checksum: FNV-1a 64 over header bytes [0, 184) followed by every byte from header_bytes to the end of the file
//                (version 1: the sections only)
//   192 = header_bytes
//   sections, in this order: gates [num_gates] x 8 u32 (type, selector_index, group_start, group_end, row,
//   num_constraints, p0, p1) | k_is [num_routed_wires] u64 | constants [num_constants][n] u64 | sigmas
//   [num_routed_wires][n] u64 | (has_witness) wires [num_wires][n] u64 | (has_witness) public_inputs
//   [num_public_inputs] u64.   n = 2^degree_bits; column-major, natural row order, canonical field elements.
// glp_circuit_file_open checks every u64 section for values < p (the kernels assume canonical inputs: a word >= p would
// give a silently wrong proof, and a consistent checksum does not exclude one) and names the first offender.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "common.h"

using namespace glp;

namespace {
constexpr char MAGIC[8] = {'G', 'L', 'P', 'C', 'I', 'R', 'C', '1'};
constexpr u32 VERSION = 2, HEADER_BYTES = 192, CHECKSUM_OFFSET = 184;

struct Header {
    char magic[8];
    u32 version, header_bytes;
    u32 scalars[14];
    u32 arity[16];
    u32 num_gates, num_public_inputs, has_witness, hasher;
    u64 digest[4];
    u64 checksum;
};
static_assert(sizeof(Header) == HEADER_BYTES, "header layout");

u64 fnv1a(const unsigned char *p, size_t len, u64 h = 0xcbf29ce484222325ull) {
    for (size_t i = 0; i < len; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}
struct Sizes { size_t gates, k_is, consts, sigmas, wires, pis, total; };
bool sizes_of(const Header &h, Sizes &s) {
    const u32 lg = h.scalars[0], nw = h.scalars[1], nr = h.scalars[2], nc = h.scalars[3];
    if (lg > 30 || nw > 4096 || nr > nw || nc > 4096 || h.num_gates > 4096 || h.num_public_inputs > (1u << 24)) return false;
    const size_t n = (size_t)1 << lg;
    s.gates = (size_t)h.num_gates * 32;
    s.k_is = (size_t)nr * 8;
    s.consts = (size_t)nc * n * 8;
    s.sigmas = (size_t)nr * n * 8;
    s.wires = h.has_witness ? (size_t)nw * n * 8 : 0;
    s.pis = h.has_witness ? (size_t)h.num_public_inputs * 8 : 0;
    s.total = HEADER_BYTES + s.gates + s.k_is + s.consts + s.sigmas + s.wires + s.pis;
    return true;
}
bool write_all(int fd, const void *p, size_t len) {
    const char *q = (const char *)p;
    while (len) {
        const ssize_t k = write(fd, q, len > ((size_t)1 << 30) ? ((size_t)1 << 30) : len);
        if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; }
        q += k; len -= (size_t)k;
    }
    return true;
}
}  // namespace

struct glp_circuit_file {
    void *map = nullptr;
    size_t len = 0;
    glp_circuit_desc desc;
    const u64 *wires = nullptr, *pis = nullptr;
};

extern "C" {

int glp_circuit_file_write(const char *path, const glp_circuit_desc *d, const uint64_t *wires, const uint64_t *public_inputs) {
    GLP_REQUIRE(path && d, "null argument");
    GLP_REQUIRE(d->gates && d->k_is && d->constants && d->sigmas, "null array in circuit description");
    GLP_REQUIRE(wires || !public_inputs, "public inputs without a witness");
    GLP_REQUIRE(!wires || public_inputs || d->num_public_inputs == 0, "witness without its public inputs");
    Header h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, MAGIC, 8);
    h.version = VERSION; h.header_bytes = HEADER_BYTES;
    const u32 sc[14] = {d->degree_bits, d->num_wires, d->num_routed_wires, d->num_constants, d->num_selectors, d->num_challenges,
                        d->quotient_degree_factor, d->num_partial_products, d->num_gate_constraints, d->rate_bits, d->cap_height,
                        d->proof_of_work_bits, d->num_query_rounds, d->num_reductions};
    memcpy(h.scalars, sc, sizeof(sc));
    GLP_REQUIRE(d->num_reductions <= 16, "more than 16 FRI reductions");
    memcpy(h.arity, d->reduction_arity_bits, sizeof(h.arity));
    h.num_gates = d->num_gates; h.num_public_inputs = d->num_public_inputs; h.has_witness = wires ? 1 : 0; h.hasher = d->hasher;
    memcpy(h.digest, d->circuit_digest, 32);
    Sizes s;
    GLP_REQUIRE(sizes_of(h, s), "circuit dimensions outside what the file format holds");
    const void *parts[6] = {d->gates, d->k_is, d->constants, d->sigmas, wires, public_inputs};
    const size_t lens[6] = {s.gates, s.k_is, s.consts, s.sigmas, s.wires, s.pis};
    static_assert(sizeof(glp_gate) == 32, "gate record is 8 x u32");
    // a writer is held to what the reader enforces: canonical field elements in every u64 section
    const char *names[6] = {"gates", "k_is", "constants", "sigmas", "wires", "public_inputs"};
    for (int i = 1; i < 6; i++) {
        if (!lens[i]) continue;
        const size_t bad = glp::first_noncanonical((const u64 *)parts[i], lens[i] / 8);
        GLP_REQUIRE(bad == lens[i] / 8, "%s[%zu] = 0x%016llx is not a canonical field element", names[i], bad, (unsigned long long)((const u64 *)parts[i])[bad]);
    }
    u64 ck = fnv1a((const unsigned char *)&h, CHECKSUM_OFFSET);
    for (int i = 0; i < 6; i++) if (lens[i]) ck = fnv1a((const unsigned char *)parts[i], lens[i], ck);
    h.checksum = ck;
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return set_error(GLP_ERR_ARG, "cannot create %s: %s", path, strerror(errno));
    bool ok = write_all(fd, &h, sizeof(h));
    for (int i = 0; i < 6 && ok; i++) if (lens[i]) ok = write_all(fd, parts[i], lens[i]);
    const int e = errno;
    if (close(fd) != 0) ok = false;
    if (!ok) { (void)unlink(path); return set_error(GLP_ERR_ARG, "write to %s failed: %s", path, strerror(e)); }
    return GLP_OK;
}

void glp_circuit_file_close(glp_circuit_file *f) {
    if (!f) return;
    if (f->map) (void)munmap(f->map, f->len);
    delete f;
}

int glp_circuit_file_open(const char *path, int verify_checksum, glp_circuit_file **out) {
    GLP_REQUIRE(path && out, "null argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return set_error(GLP_ERR_ARG, "cannot open %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < HEADER_BYTES) { (void)close(fd); return set_error(GLP_ERR_ARG, "%s: shorter than a circuit-file header", path); }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);   // GB-sized circuits: map, do not copy
    (void)close(fd);
    if (m == MAP_FAILED) return set_error(GLP_ERR_ARG, "mmap %s: %s", path, strerror(errno));
    std::unique_ptr<glp_circuit_file, void (*)(glp_circuit_file *)> f(new glp_circuit_file(), glp_circuit_file_close);
    f->map = m; f->len = (size_t)st.st_size;
    Header h;
    memcpy(&h, m, sizeof(h));
    GLP_REQUIRE(memcmp(h.magic, MAGIC, 8) == 0, "%s: not a circuit file (bad magic)", path);
    if (h.version != VERSION && h.version != 1) return set_error(GLP_ERR_UNSUPPORTED, "%s: circuit-file version %u, this build reads versions 1 and %u", path, h.version, VERSION);
    GLP_REQUIRE(h.header_bytes == HEADER_BYTES && h.has_witness <= 1 && h.scalars[13] <= 16 && h.hasher <= 1, "%s: malformed header", path);
    Sizes s;
    GLP_REQUIRE(sizes_of(h, s), "%s: circuit dimensions out of range", path);
    GLP_REQUIRE(s.total == f->len, "%s: %zu bytes, the header describes %zu (truncated or padded file)", path, f->len, s.total);
    const unsigned char *base = (const unsigned char *)m;
    if (verify_checksum) {
        const u64 seed = h.version >= 2 ? fnv1a(base, CHECKSUM_OFFSET) : 0xcbf29ce484222325ull;      // version 2 covers the header too
        const u64 ck = fnv1a(base + HEADER_BYTES, f->len - HEADER_BYTES, seed);
        GLP_REQUIRE(ck == h.checksum, "%s: checksum mismatch (file corrupted)", path);
    }
    glp_circuit_desc &d = f->desc;
    memset(&d, 0, sizeof(d));
    d.degree_bits = h.scalars[0]; d.num_wires = h.scalars[1]; d.num_routed_wires = h.scalars[2]; d.num_constants = h.scalars[3];
    d.num_selectors = h.scalars[4]; d.num_challenges = h.scalars[5]; d.quotient_degree_factor = h.scalars[6];
    d.num_partial_products = h.scalars[7]; d.num_gate_constraints = h.scalars[8]; d.rate_bits = h.scalars[9]; d.cap_height = h.scalars[10];
    d.proof_of_work_bits = h.scalars[11]; d.num_query_rounds = h.scalars[12]; d.num_reductions = h.scalars[13];
    memcpy(d.reduction_arity_bits, h.arity, sizeof(h.arity));
    d.num_gates = h.num_gates; d.num_public_inputs = h.num_public_inputs; d.hasher = h.hasher;
    memcpy(d.circuit_digest, h.digest, 32);
    size_t o = HEADER_BYTES;
    d.gates = (const glp_gate *)(base + o); o += s.gates;
    d.k_is = (const u64 *)(base + o); o += s.k_is;
    d.constants = (const u64 *)(base + o); o += s.consts;
    d.sigmas = (const u64 *)(base + o); o += s.sigmas;
    if (h.has_witness) { f->wires = (const u64 *)(base + o); o += s.wires; f->pis = (const u64 *)(base + o); }
    {
        const u64 *sec[5] = {d.k_is, d.constants, d.sigmas, f->wires, f->pis};
        const size_t cnt[5] = {s.k_is / 8, s.consts / 8, s.sigmas / 8, s.wires / 8, s.pis / 8};
        const char *names[5] = {"k_is", "constants", "sigmas", "wires", "public_inputs"};
        for (int i = 0; i < 5; i++) {
            if (!cnt[i]) continue;
            const size_t bad = glp::first_noncanonical(sec[i], cnt[i]);
            GLP_REQUIRE(bad == cnt[i], "%s: %s[%zu] = 0x%016llx is not a canonical field element (>= p)", path, names[i], bad, (unsigned long long)sec[i][bad]);
        }
    }
    *out = f.release();
    return GLP_OK;
}

const glp_circuit_desc *glp_circuit_file_desc(const glp_circuit_file *f) { return f ? &f->desc : nullptr; }
const uint64_t *glp_circuit_file_wires(const glp_circuit_file *f) { return f ? f->wires : nullptr; }
const uint64_t *glp_circuit_file_public_inputs(const glp_circuit_file *f) { return f ? f->pis : nullptr; }

}  // extern "C"
