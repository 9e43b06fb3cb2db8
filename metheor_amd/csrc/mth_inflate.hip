// mth_inflate.hip -- BGZF (RFC 1951 DEFLATE) block inflate and per-block record walk on the device
// (the step before the record decode; replaces bamutil.rs:4-11's htslib reader thread).
//
// BGZF = independent <= 64-KiB gzip members, each one raw DEFLATE stream: embarrassingly parallel across blocks,
// strictly serial inside one.  Mapping: ONE WAVE per BGZF block.  The decoder state (bit buffer, input cursor,
// output cursor) is wave-uniform, so the symbol loop is effectively one decoder per wave; the lanes are used for
// what is parallel -- filling the Huffman lookup tables (lane = symbol) and copying matches (lane = byte).
// Huffman decode: a direct lookup table on the low ROOT bits (codes are read LSB-first, so the index is the
// bit-reversed code) for codes up to ROOT bits, canonical bit-by-bit decode (count / first-code walk) for the few
// longer ones.  Output goes straight to HBM; back-references read it back (the window of a block is the block).
// A block is checked against its ISIZE; any inconsistency sets ERRB_FORMAT.
//
// k_block_walk: BAM writers built on htslib flush the BGZF block before a record that would not fit, so records
// never straddle blocks and every block starts at a record boundary: one thread per block follows the block_size
// chain, counting / writing record offsets, and VERIFIES that the chain ends exactly at the block's end.  A file for
// which that does not hold sets ERRB_UNALIGNED and the caller takes the host walk instead.
#include "mth_ctx.h"

namespace mth {

// Reads of a CORRUPT last block can run past the staged file bytes: garbage keeps decoding until the output bound (isize <= 64 KiB)
// stops it.  A literal costs at most 15 bits (the direct-lookup width LROOT is not the code length limit: DEFLATE codes go up to 15
// bits) = 120 KiB of input for 64 Ki literals; a match costs at most 15 + 5 + 15 + 13 bits and yields >= 3 bytes: less per output
// byte.  The staging buffers carry 160 KiB of zeroed padding, so the bit reader's scalar loads always stay inside the allocation
// and read defined bytes (the CRC / ISIZE checks then flag the block).
constexpr size_t INF_FILE_PAD = 160 * 1024;

constexpr int LROOT = 10, DROOT = 9;          // direct-lookup bits of the literal/length and distance tables
struct InflArgs {
    const uint8_t *file;                      // the compressed file, padded by >= 16 readable bytes
    const uint64_t *coff;                     // per block: offset of the DEFLATE payload in the file
    const uint32_t *csize, *isize;            // payload bytes / inflated bytes
    const uint64_t *uoff;                     // offset of the block's output in the inflated stream
    uint32_t n_blocks;
    uint8_t *out;
    uint32_t *err;
    int fast_literals;                         // the hand-written literal loop (METHEOR_INFLATE_ASM=0 turns it off: A/B)
};

__constant__ uint8_t c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// one Huffman code: lens[0..n) -> direct table (entry = sym << 4 | len, 0 = not a short code) + canonical arrays
struct Huff {
    uint16_t *tab;        // 1 << root entries
    uint16_t *sorted;     // symbols ordered by (length, symbol)
    uint16_t *count;      // [16] codes per length
    int root;
    uint16_t *tmp;        // [32] builder scratch (next code / sorted offset per length)
};

// wave-cooperative build; returns false for an over-subscribed code (incomplete codes are accepted as zlib does for
// the single-distance-code case; an unused entry decodes as an error later)
__device__ bool huff_build(const Huff &h, const uint8_t *lens, int n, uint16_t *code_of /* scratch, n entries */, bool flag_literals = false) {
    const int lane = threadIdx.x & 63;
    if (lane < 16) h.count[lane] = 0;
    for (int i = lane; i < (1 << h.root); i += 64) h.tab[i] = 0;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {                  // canonical codes (RFC 1951 3.2.2); per-length arrays in LDS (dynamic indexing)
        uint16_t *cnt = h.count, *next = h.tmp, *offs = h.tmp + 16;
        for (int s = 0; s < n; ++s) cnt[lens[s]]++;
        cnt[0] = 0;
        int left = 1;
        bool over = false;
        for (int l = 1; l < 16; ++l) { left <<= 1; left -= cnt[l]; if (left < 0) over = true; }
        uint32_t code = 0;
        uint16_t o = 0;
        next[0] = 0; offs[0] = 0;
        for (int l = 1; l < 16; ++l) { code = (code + cnt[l - 1]) << 1; next[l] = (uint16_t)code; offs[l] = o; o += cnt[l]; }
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (l) { code_of[s] = next[l]++; h.sorted[offs[l]++] = (uint16_t)s; } else code_of[s] = 0;
        }
        cnt[0] = over ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (h.count[0]) return false;
    // direct table: lane = symbol; a code of length l <= root owns every index whose low l bits are its reversal
    for (int s = lane; s < n; s += 64) {
        const int l = lens[s];
        if (l == 0 || l > h.root) continue;
        const uint32_t rev = __builtin_bitreverse32((uint32_t)code_of[s]) >> (32 - l);
        const uint16_t e = (uint16_t)((s << 4) | l | ((flag_literals && s < 256) ? 0x8000 : 0));   // bit 15: a literal (the asm loop tests it)
        for (uint32_t k = rev; k < (1u << h.root); k += (1u << l)) h.tab[k] = e;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return true;
}

// The bit reader.  Wave-uniform by construction: every field lives in scalar registers and the input is read with SCALAR
// loads of aligned dwords (the file bytes are not written by this launch), so a refill neither occupies a vector lane nor
// waits for the literal stores in flight (a vector load's s_waitcnt vmcnt does).
// (the decoder state IS wave-uniform, but values that came through LDS or a vector load look divergent to the compiler:
// readfirstlane states the fact at the scalar-asm boundaries)
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t uni64(uint64_t x) { return (uint64_t)uni((uint32_t)x) | ((uint64_t)uni((uint32_t)(x >> 32)) << 32); }
template <class T> __device__ __forceinline__ T *uni_ptr(T *p) { return reinterpret_cast<T *>(uni64(reinterpret_cast<uint64_t>(p))); }
__device__ __forceinline__ uint32_t sload_dword(const uint8_t *p) {   // p: wave-uniform, 4-byte aligned
    uint32_t w;
    const uint8_t *q = uni_ptr(p);
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(q) : "memory");
#ifndef MTH_INFLATE_DIVERGENT_STATE
    w = uni(w);                          // (an asm result counts as divergent: say it is not, or the whole decoder moves to the vector unit)
#endif
    return w;
}
struct Bits {
    const uint8_t *file;         // the compressed bytes (a kernel argument: loads through file + offset stay global, uniform loads --
                                 // a pointer that came back from the scalar asm would be a flat one, and flat loads count as divergent)
    uint64_t in, end;            // byte offsets in file: next aligned dword to load; first byte past the payload (reads run a few bytes beyond: the file is padded)
    uint64_t bb;
    int bc;
    __device__ __forceinline__ void seek(uint64_t p) {          // start reading at byte offset p
        const uint32_t skip = ((uint32_t)(reinterpret_cast<uintptr_t>(file) + p) & 3u) * 8u;
        in = p - (skip >> 3);
        const uint32_t w = sload_dword(file + in);
        in += 4;
        bb = w >> skip; bc = 32 - (int)skip;
    }
    __device__ __forceinline__ void refill() {
        if (bc < 32) {
            const uint32_t w = sload_dword(file + in);
            bb |= (uint64_t)w << bc;
            in += 4; bc += 32;
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)bb & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n) { bb >>= n; bc -= n; }
    __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
    // bytes consumed so far (whole bytes of the bit buffer given back), as an offset in file
    __device__ __forceinline__ uint64_t cursor() const { return in - (uint64_t)(bc >> 3); }
};

// A run of literals in hand-written ISA: while the next code is a direct-table hit for a literal (bit 15 of the entry), decode
// it, store it and go on -- three per trip around the loop (a refill leaves >= 32 bits, a table hit takes <= 10), refilling
// from scalar memory as it goes; table index, literal test and code length are worked out on the vector side (the scalar unit
// is the busy one): ~5 scalar + 8 vector instructions per literal against the ~30 + 15 the compiler made of the
// general symbol loop; the scalar unit, one per CU, is what k_inflate is bound by (PMC: 10.8 G scalar / 5.8 G vector
// instructions per 13 441 blocks before, profiles/r02_e2e.md).  The output bound is checked where the bits are refilled: at
// most 63 + 2 literals can follow a refill, so the loop only runs while 72 bytes of room are left and the block's last bytes
// go through the general loop.  Leaves at anything else (length code, end of block, long code) with the state as the general
// loop expects it.
#define MTH_LIT_STEP                                     \
    "v_mov_b32 %[vt], s40\n\t"                           \
    "v_and_b32 %[vt], 0x3ff, %[vt]\n\t"                  \
    "v_lshl_add_u32 %[vt], %[vt], 1, %[vlt]\n\t"         \
    "ds_read_u16 %[ve], %[vt]\n\t"                       \
    "s_waitcnt lgkmcnt(0)\n\t"                           \
    "v_cmp_gt_u32 vcc, 0x8000, %[ve]\n\t"                \
    "v_and_b32 %[vt], 15, %[ve]\n\t"                     \
    "s_cbranch_vccnz L_exit_%=\n\t"                      \
    "v_readfirstlane_b32 %[e], %[vt]\n\t"                \
    "s_lshr_b64 s[40:41], s[40:41], %[e]\n\t"            \
    "s_sub_u32 %[bc], %[bc], %[e]\n\t"                   \
    "v_lshrrev_b32 %[ve], 4, %[ve]\n\t"                  \
    "global_store_byte %[vpos], %[ve], %[out]\n\t"       \
    "v_add_u32 %[vpos], 1, %[vpos]\n\t"
__device__ __forceinline__ void literal_run(Bits &b, uint32_t &pos, const uint32_t isize, uint8_t *out0, const uint32_t ltab_lds) {
    // The loop checks the output bound where it refills; entered with >= 32 bits in hand it would store up to 63 + 2 literals
    // before its first check: same bound here, once per call (a corrupt stream must not write into the next block's output).
    if (pos + 72u > isize) return;
    uint64_t bb = uni64(b.bb), tmp;
    const uint64_t base = uni64(reinterpret_cast<uint64_t>(b.file));
    uint64_t in = base + uni64(b.in);                   // the address as an integer: scalar registers s[42:43]
    int32_t bc = (int32_t)uni((uint32_t)b.bc);
    uint32_t t, e, vt, ve;
    uint32_t vpos = pos;
    const uint32_t u_isize = uni(isize), u_lt = uni(ltab_lds);
    uint8_t *const u_out = uni_ptr(out0);
    asm volatile(
        "L_top_%=:\n\t"
        "s_cmp_lt_i32 %[bc], 32\n\t"
        "s_cbranch_scc0 L_go_%=\n\t"
        "v_readfirstlane_b32 %[t], %[vpos]\n\t"
        "s_add_u32 %[t], %[t], 72\n\t"
        "s_cmp_gt_u32 %[t], %[isize]\n\t"
        "s_cbranch_scc1 L_exit_%=\n\t"
        "s_load_dword s44, s[42:43], 0x0\n\t"
        "s_add_u32 s42, s42, 4\n\t"
        "s_addc_u32 s43, s43, 0\n\t"
        "s_mov_b32 s45, 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_lshl_b64 s[44:45], s[44:45], %[bc]\n\t"
        "s_or_b64 s[40:41], s[40:41], s[44:45]\n\t"
        "s_add_u32 %[bc], %[bc], 32\n\t"
        "L_go_%=:\n\t"
        MTH_LIT_STEP MTH_LIT_STEP MTH_LIT_STEP
        "s_branch L_top_%=\n\t"
        "L_exit_%=:\n\t"
        : "+{s[40:41]}"(bb), "+{s[42:43]}"(in), [bc] "+s"(bc), [vpos] "+v"(vpos), "=&{s[44:45]}"(tmp), [t] "=&s"(t), [e] "=&s"(e),
          [vt] "=&v"(vt), [ve] "=&v"(ve)
        : [vlt] "v"(u_lt), [isize] "s"(u_isize), [out] "s"(u_out)
        : "memory", "scc", "vcc");
#ifndef MTH_INFLATE_DIVERGENT_STATE
    b.bb = uni64(bb); b.in = uni64(in) - base; b.bc = (int)uni((uint32_t)bc); pos = uni(vpos);
#else
    b.bb = bb; b.in = in - base; b.bc = bc; pos = vpos;
#endif
}
#undef MTH_LIT_STEP

// decode one symbol; -1 on an invalid code
__device__ __forceinline__ int huff_decode(const Huff &h, Bits &b) {
    const uint32_t e = h.tab[b.peek(h.root)];
    if (e) { b.drop((int)(e & 15u)); return (int)((e >> 4) & 0x7ffu); }
    // longer than root bits (or invalid): canonical walk, one bit at a time (RFC 1951 3.2.2; MSB of the code first)
    int code = 0, first = 0, index = 0;
#pragma unroll 1                                       // rare path: unrolled 15 deep it tripled the kernel and cost it a wave per SIMD
    for (int l = 1; l < 16; ++l) {
        code |= (int)b.take(1);
        const int c = h.count[l];
        if (code - c < first) return h.sorted[index + (code - first)];
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}

// The symbols of one DEFLATE block (RFC 1951 3.2.3-3.2.5) up to its end-of-block code; true = the stream is invalid.
// Early returns instead of loop-carried error flags: the flags were most of what the compiler's version spent per symbol.
__device__ __forceinline__ bool inflate_symbols(Bits &b, const Huff &HL, const Huff &HD, uint32_t &pos, const uint32_t isize, uint8_t *out0,
                                                const uint32_t ltab_lds, const bool fast_literals, const int lane, const uint32_t *s_lbx,
                                                const uint32_t *s_dbx) {
    for (;;) {
        if (fast_literals) literal_run(b, pos, isize, out0, ltab_lds);
        b.refill();
        const int sym = huff_decode(HL, b);
        if (sym < 0) return true;
        if (sym < 256) {
            if (pos >= isize) return true;
            out0[pos] = (uint8_t)sym;          // every lane, same address, same value: no exec-mask juggling for lane 0
            ++pos;
            continue;
        }
        if (sym == 256) return false;
        if (sym > 285) return true;
        const int li = sym - 257;
        // base and extra bits of a length / distance symbol (RFC 1951 3.2.5) from two small LDS tables built at kernel start:
        // by arithmetic they were ~25 scalar instructions per match, on the unit the kernel is bound by
        const uint32_t lbx = s_lbx[li];
        const uint32_t len = (lbx & 0xffffu) + b.take((int)(lbx >> 16));
        b.refill();
        const int ds = huff_decode(HD, b);
        if (ds < 0 || ds > 29) return true;
        const uint32_t dbx = s_dbx[ds];
        const uint32_t dist = (dbx & 0xffffu) + b.take((int)(dbx >> 16));
        if (dist > pos || pos + len > isize) return true;
        // the bytes written so far must be visible to the loads below (same wave, but loads and stores return
        // out of order with respect to each other)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (dist >= len) {                              // no overlap: all lanes copy at once
            for (uint32_t k = lane; k < len; k += 64) out0[pos + k] = out0[pos - dist + k];
        } else {                                        // overlapping run: the source repeats with period dist
            for (uint32_t k = lane; k < len; k += 64) out0[pos + k] = out0[pos - dist + (k % dist)];
        }
        pos += len;
    }
}

__global__ __launch_bounds__(64, 8) void k_inflate(const InflArgs a) {
    __shared__ uint16_t s_ltab[1 << LROOT], s_dtab[1 << DROOT];
    __shared__ uint16_t s_lsorted[288], s_dsorted[32], s_lcount[16], s_dcount[16], s_code[320];
    __shared__ uint8_t s_lens[320];
    __shared__ uint16_t s_tmp[32];
    __shared__ uint32_t s_lbx[32], s_dbx[32];     // length / distance symbols: base | extra bits << 16 (RFC 1951 3.2.5)
    const uint32_t blk = blockIdx.x;
    if (blk >= a.n_blocks) return;
    const int lane = threadIdx.x;
    if (lane < 29) {                               // from the 9th length symbol on, four symbols per extra-bit count
        const uint32_t li = (uint32_t)lane, lx = (li < 8 || li == 28) ? 0u : (li >> 2) - 1u;
        s_lbx[lane] = (li < 8 ? li + 3u : (li == 28 ? 258u : ((4u + (li & 3u)) << lx) + 3u)) | (lx << 16);
    }
    if (lane < 30) {                               // from the 5th distance symbol on, two per extra-bit count
        const uint32_t ds = (uint32_t)lane, dx = ds < 4 ? 0u : (ds >> 1) - 1u;
        s_dbx[lane] = (ds < 4 ? ds + 1u : ((2u + (ds & 1u)) << dx) + 1u) | (dx << 16);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint8_t *const out0 = a.out + a.uoff[blk];
    const uint32_t isize = a.isize[blk];
    Bits b;
    b.file = a.file;
    b.end = a.coff[blk] + a.csize[blk];
    b.seek(a.coff[blk]);
    const uint32_t ltab_lds = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint16_t *)s_ltab);
    const Huff HL{s_ltab, s_lsorted, s_lcount, LROOT, s_tmp}, HD{s_dtab, s_dsorted, s_dcount, DROOT, s_tmp};
    uint32_t pos = 0;
    bool bad = false, last = false;
    while (!last && !bad) {
        b.refill();
        last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (type == 0) {                                   // stored
            b.drop(b.bc & 7);                              // to the byte boundary
            const uint64_t po = b.cursor();
            const uint8_t *p = a.file + po;
            const uint32_t len = (uint32_t)p[0] | ((uint32_t)p[1] << 8), nlen = (uint32_t)p[2] | ((uint32_t)p[3] << 8);
            p += 4;
            if ((len ^ nlen) != 0xffffu || pos + len > isize || po + 4 + len > b.end) { bad = true; break; }
            for (uint32_t i = lane; i < len; i += 64) out0[pos + i] = p[i];
            pos += len;
            b.seek(po + 4 + len);
            continue;
        }
        if (type == 3) { bad = true; break; }
        if (type == 1) {                                   // fixed codes (RFC 1951 3.2.6)
            for (int s = lane; s < 288; s += 64) s_lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (!huff_build(HL, s_lens, 288, s_code, true)) { bad = true; break; }
            for (int s = lane; s < 30; s += 64) s_lens[s] = 5;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (!huff_build(HD, s_lens, 30, s_code)) { bad = true; break; }
        } else {                                           // dynamic codes (3.2.7)
            b.refill();
            const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
            if (hlit > 286 || hdist > 30) { bad = true; break; }
            if (lane < 19) s_lens[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int i = 0; i < hclen; ++i) {
                b.refill();
                const uint32_t v = b.take(3);
                if (lane == 0) s_lens[c_clen_order[i]] = (uint8_t)v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // the code-length code reuses the distance table's storage (root 7 <= DROOT)
            const Huff HC{s_dtab, s_dsorted, s_dcount, 7, s_tmp};
            if (!huff_build(HC, s_lens, 19, s_code)) { bad = true; break; }
            int i = 0, prev = 0;
            const int total = hlit + hdist;
            while (i < total) {
                b.refill();
                const int sym = huff_decode(HC, b);
                if (sym < 0) { bad = true; break; }
                int rep = 1, val = sym;
                if (sym == 16) { if (i == 0) { bad = true; break; } val = prev; rep = 3 + (int)b.take(2); }
                else if (sym == 17) { val = 0; rep = 3 + (int)b.take(3); }
                else if (sym == 18) { val = 0; rep = 11 + (int)b.take(7); }
                if (i + rep > total) { bad = true; break; }
                // lens of the literal/length code then of the distance code, one array (s_lens is 320 long)
                for (int k = lane; k < rep; k += 64) s_lens[i + k] = (uint8_t)val;   // into a second region below
                i += rep; prev = val;
            }
            if (bad) break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (s_lens[256] == 0) { bad = true; break; }   // no end-of-block code
            // the distance lengths follow the literal/length ones: build the distance code first (it lives in s_dtab,
            // which the code-length code no longer needs), from a copy at the front of a scratch region
            if (!huff_build(HD, s_lens + hlit, hdist, s_code)) { bad = true; break; }
            if (!huff_build(HL, s_lens, hlit, s_code, true)) { bad = true; break; }
        }
        // ---- symbols ----
        if (inflate_symbols(b, HL, HD, pos, isize, out0, ltab_lds, a.fast_literals != 0, lane, s_lbx, s_dbx)) { bad = true; break; }
    }
    if (!bad && (pos != isize || b.cursor() > b.end)) bad = true;
    if (bad && lane == 0) atomicOr(a.err, (uint32_t)ERRB_FORMAT);
}


// ---- CRC32 of the inflated blocks (gzip trailer; htslib and the host reader verify it, so does this path) ----------
// One wave per block.  Raw CRC (init 0, no final xor) is linear and ignores leading zeros, so the block is cut into 64
// chunks of 1 KiB aligned to its END (the first chunks of a short block are empty); lane = chunk, slicing-by-4 table
// CRC from LDS tables, then a 6-level tree R(A||B) = shift_|B|(R(A)) ^ R(B) with precomputed GF(2) operators for
// "append 2^k zero bytes" (crc_mat[k][32]).  The standard CRC is ~(R0(data) ^ shift_L(0xffffffff)).
struct CrcArgs {
    const uint8_t *raw;             // inflated stream
    const uint8_t *file;            // compressed file (the stored CRC follows each payload)
    const uint64_t *coff, *uoff;
    const uint32_t *csize, *isize;
    const uint32_t *mat;            // [17][32]: operator for 2^k zero bytes, k = 0..16
    uint32_t n_blocks;
    uint32_t *err;
};
__device__ __forceinline__ uint32_t gf2_apply(const uint32_t *__restrict__ m, uint32_t v) {
    uint32_t s = 0;
#pragma unroll 4
    for (int i = 0; i < 32; ++i) s ^= ((v >> i) & 1u) ? m[i] : 0u;
    return s;
}
__global__ __launch_bounds__(64) void k_crc32(const CrcArgs a) {
    __shared__ uint32_t tab[4][256];          // slicing-by-4: tab[j][b] = CRC of byte b followed by j zero bytes
    __shared__ uint32_t smat[17 * 32];
    const uint32_t blk = blockIdx.x;
    if (blk >= a.n_blocks) return;
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) {
        uint32_t c = (uint32_t)i;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        tab[0][i] = c;
    }
    for (int i = lane; i < 17 * 32; i += 64) smat[i] = a.mat[i];
    __syncthreads();
    for (int i = lane; i < 256; i += 64) {
        const uint32_t c1 = (tab[0][i] >> 8) ^ tab[0][tab[0][i] & 0xffu];
        const uint32_t c2 = (c1 >> 8) ^ tab[0][c1 & 0xffu];
        tab[1][i] = c1; tab[2][i] = c2; tab[3][i] = (c2 >> 8) ^ tab[0][c2 & 0xffu];
    }
    __syncthreads();
    const uint32_t L = a.isize[blk];
    const uint8_t *p = a.raw + a.uoff[blk];
    const int64_t beg = (int64_t)L - (int64_t)(64 - lane) * 1024;
    uint32_t c = 0;
    int64_t i = beg < 0 ? 0 : beg;
    const int64_t end = beg + 1024;                          // <= 0 for the empty leading chunks of a short block
    for (; i < end && ((end - i) & 3); ++i) c = tab[0][(c ^ p[i]) & 0xffu] ^ (c >> 8);
    auto step4 = [&](uint32_t w) {                          // four bytes: four independent lookups
        c ^= w;
        c = tab[3][c & 0xffu] ^ tab[2][(c >> 8) & 0xffu] ^ tab[1][(c >> 16) & 0xffu] ^ tab[0][c >> 24];
    };
    // sixteen bytes per load: the lanes of a wave read 64 different cache lines (their chunks lie 1 KiB apart) and 28 waves
    // per CU do not fit L1, so every load is an L2 round trip -- a quarter as many of them
    typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
    for (; i + 16 <= end; i += 16) {
        const u32x4_a1 w = *reinterpret_cast<const u32x4_a1 *>(p + i);
        step4(w.x); step4(w.y); step4(w.z); step4(w.w);
    }
    for (; i < end; i += 4) {
        uint32_t w;
        __builtin_memcpy(&w, p + i, 4);
        step4(w);
    }
    // tree: at level k lanes are grouped in runs of 2^(k+1); the left half's CRC is shifted by the right half's length
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t other = (uint32_t)__shfl_xor((int)c, 1 << k, 64);
        const bool right = (lane >> k) & 1;                      // this lane holds the right half of its pair
        const uint32_t left_crc = right ? other : c, right_crc = right ? c : other;
        c = gf2_apply(smat + (10 + k) * 32, left_crc) ^ right_crc;   // right half is 1024 << k bytes long
    }
    if (lane == 0) {
        uint32_t init = 0xffffffffu;                              // shift_L(0xffffffff): binary decomposition of L
        for (int k = 0; k < 17; ++k) if ((L >> k) & 1u) init = gf2_apply(smat + k * 32, init);
        const uint32_t crc = ~(c ^ init);
        uint32_t want;
        __builtin_memcpy(&want, a.file + a.coff[blk] + a.csize[blk], 4);
        if (crc != want) atomicOr(a.err, (uint32_t)ERRB_CRC);
    }
}

// ---- per-block record walk ------------------------------------------------------------------------------
struct WalkBArgs {
    const uint8_t *raw;            // inflated stream
    const uint64_t *uoff;          // per block: offset of its bytes in raw
    const uint32_t *isize;
    uint32_t n_blocks;
    uint64_t first_byte;           // records start here (end of the BAM header)
    uint32_t *cnt;                 // pass 1: records per block
    const unsigned long long *base;   // pass 2: exclusive scan of cnt
    uint64_t *rec_off;             // pass 2: record offsets (+ closing offset written by the last block)
    uint64_t total_bytes;
    uint32_t *err;
};
template <bool FILL>
__global__ __launch_bounds__(256) void k_block_walk(const WalkBArgs a) {
    const uint32_t blk = blockIdx.x * 256 + threadIdx.x;
    if (blk >= a.n_blocks) return;
    const uint64_t b0 = a.uoff[blk], b1 = b0 + a.isize[blk];
    uint32_t n = 0;
    if (b1 > a.first_byte) {
        uint64_t p = b0 < a.first_byte ? a.first_byte : b0;
        unsigned long long w = FILL ? a.base[blk] : 0ull;
        while (p + 4 <= b1) {
            uint32_t bs;
            __builtin_memcpy(&bs, a.raw + p, 4);
            if (bs < 32 || p + 4 + (uint64_t)bs > b1) break;
            if (FILL) a.rec_off[w++] = p;
            ++n;
            p += 4 + (uint64_t)bs;
        }
        if (p != b1) atomicOr(a.err, (uint32_t)ERRB_UNALIGNED);      // a record straddles the block (or garbage)
    }
    if (!FILL) a.cnt[blk] = n;
    if (FILL && blk == a.n_blocks - 1) a.rec_off[a.base[blk] + n] = a.total_bytes;
}

}  // namespace mth

using namespace mth;

// GF(2) operators for appending 2^k zero bytes to a raw CRC-32 (zlib's crc32_combine construction)
static void crc_zero_operators(uint32_t (*mat)[32]) {
    auto times = [](const uint32_t *m, uint32_t v) { uint32_t s = 0; for (int i = 0; v; v >>= 1, ++i) if (v & 1u) s ^= m[i]; return s; };
    uint32_t bit[32], t[32];
    bit[0] = 0xedb88320u;                               // one zero BIT
    for (int n = 1; n < 32; ++n) bit[n] = 1u << (n - 1);
    auto square = [&](uint32_t *dst, const uint32_t *src) { for (int n = 0; n < 32; ++n) dst[n] = times(src, src[n]); };
    square(t, bit); square(bit, t); square(t, bit);     // 2, 4, 8 bits: t = one zero byte
    for (int n = 0; n < 32; ++n) mat[0][n] = t[n];
    for (int k = 1; k < 17; ++k) square(mat[k], mat[k - 1]);
}

// (Splitting a pageable host-to-device copy over 4 / 8 host threads that enqueue slices into the same stream was measured:
// no gain at 4, slower at 8 -- profiles/r02_e2e.md.)
// stage the file bytes + block table and launch the inflate; on return d_uoff / d_isize point at the device table
static int inflate_blocks(mth_ctx *ctx, const void *file, uint64_t n_bytes, const uint64_t *coff, const uint32_t *csize,
                          const uint32_t *isize, uint64_t n_blocks, uint64_t &total, uint64_t *&d_uoff, uint32_t *&d_isize) {
    if (n_blocks >= (1ull << 31)) return fail(ctx, MTH_ERR_CAPACITY, "too many BGZF blocks in one call: split the file");
    MTH_ENTER(ctx);
    hipStream_t s = ctx->stream;
    const size_t nb = (size_t)n_blocks;
    // output offsets of the blocks + validation of the table against the byte range given
    std::vector<uint64_t> uoff(nb + 1, 0);
    for (size_t i = 0; i < nb; ++i) {
        if (coff[i] + (uint64_t)csize[i] > n_bytes || isize[i] > 65536u) return fail(ctx, MTH_ERR_INVALID, "BGZF block table does not fit the file bytes");
        uoff[i + 1] = uoff[i] + isize[i];
    }
    total = uoff[nb];
    // stage: compressed bytes (padded: the bit reader loads whole words), the table, the inflated stream
    // the helper thread of the previous call (copying THIS call's bytes, if they were announced) has to be done first
    if (ctx->stage_thread.joinable()) {
        ctx->stage_thread.join();
        if (ctx->stage_rc != MTH_OK) return fail(ctx, MTH_ERR_HIP, "staging the next chunk's file bytes failed");
    }
    const bool staged = ctx->staged_src && ctx->staged_src == file && ctx->staged_bytes == n_bytes;
    if (staged) {
        // copied on the side stream while the previous chunk was in flight: take that buffer
        std::swap(ctx->inf_file, ctx->inf_file2);
        MTH_HIP(ctx, hipStreamWaitEvent(s, ctx->staged_ev, 0));
    } else {
        MTH_HIP(ctx, ctx->inf_file.reserve((size_t)n_bytes + INF_FILE_PAD, s));
    }
    ctx->staged_src = nullptr;
    MTH_HIP(ctx, ctx->inf_tab.reserve(nb * 24 + 64, s));
    MTH_HIP(ctx, ctx->inf_raw.reserve((size_t)total + 64, s));
    // METHEOR_COPY_PIECE_MB=k (default 0 = off): an in-line copy (the first chunk of a file, or every chunk under METHEOR_NO_STAGE)
    // of two pieces or more travels on its own stream piece by piece, and the blocks of a piece are inflated as soon as the piece
    // has landed.  Built to hide the first chunk's inflate behind its copy; measured slower (profiles/r02_e2e.md, last section):
    // a pageable copy pins, moves and unpins inside the call, ~2.5 ms of that per call is not overlapped with anything, and the
    // runtime pipelines one large copy better than the caller can pipeline several small ones.  Kept as a switch, parity-tested.
    static const size_t piece_bytes = [] { const char *e = getenv("METHEOR_COPY_PIECE_MB"); const long k = e ? atol(e) : 0; return k > 0 ? (size_t)k << 20 : (size_t)0; }();
    const bool pieced = !staged && piece_bytes && n_bytes >= 2 * (uint64_t)piece_bytes && nb >= 2;
    if (!staged && !pieced) {
        // (copying through own page-locked pieces from 2-8 worker threads was measured: no faster than the runtime's pageable path,
        // profiles/r02_e2e.md -- the page-cache reads bound both)
        if (n_bytes) MTH_HIP(ctx, hipMemcpyAsync(ctx->inf_file.p, file, (size_t)n_bytes, hipMemcpyHostToDevice, s));
        MTH_HIP(ctx, hipMemsetAsync(static_cast<uint8_t *>(ctx->inf_file.p) + n_bytes, 0, INF_FILE_PAD, s));
    }
    // (before the inflate is launched: allocating it afterwards synchronised with the inflate and delayed the first CRC launch by ~7 ms)
    if (!ctx->crc_mat.p) {
        uint32_t mat[17][32];
        crc_zero_operators(mat);
        MTH_HIP(ctx, ctx->crc_mat.reserve(sizeof mat, s));
        MTH_HIP(ctx, hipMemcpyAsync(ctx->crc_mat.p, mat, sizeof mat, hipMemcpyHostToDevice, s));
        MTH_HIP(ctx, hipStreamSynchronize(s));            // mat is a local
    }
    // the chunk announced by mth_bgzf_stage: its bytes travel on the side stream, from a helper thread (a pageable copy
    // blocks its caller), while this call's kernels run.  The thread owns inf_file2 / copy_stream / staged_* until joined.
    if (ctx->next_src && ctx->next_bytes) {
        if (!ctx->copy_stream) {
            MTH_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            MTH_HIP(ctx, hipEventCreateWithFlags(&ctx->staged_ev, hipEventDisableTiming));
        }
        const void *src = ctx->next_src;
        const uint64_t nbytes = ctx->next_bytes;
        ctx->next_src = nullptr;
        ctx->stage_rc = MTH_OK;
        ctx->stage_thread = std::thread([ctx, src, nbytes] {
            auto ok = [&](hipError_t e) { if (e != hipSuccess) ctx->stage_rc = MTH_ERR_HIP; return e == hipSuccess; };
            if (!ok(hipSetDevice(ctx->device))) return;
            if (!ok(ctx->inf_file2.reserve((size_t)nbytes + INF_FILE_PAD, ctx->copy_stream))) return;
            if (!ok(hipMemcpyAsync(ctx->inf_file2.p, src, (size_t)nbytes, hipMemcpyHostToDevice, ctx->copy_stream))) return;
            if (!ok(hipMemsetAsync(static_cast<uint8_t *>(ctx->inf_file2.p) + nbytes, 0, INF_FILE_PAD, ctx->copy_stream))) return;
            if (!ok(hipEventRecord(ctx->staged_ev, ctx->copy_stream))) return;
            ctx->staged_src = src; ctx->staged_bytes = nbytes;
        });
    }
    uint8_t *tab = static_cast<uint8_t *>(ctx->inf_tab.p);
    uint64_t *d_coff = reinterpret_cast<uint64_t *>(tab);
    d_uoff = reinterpret_cast<uint64_t *>(tab + nb * 8);
    uint32_t *d_csize = reinterpret_cast<uint32_t *>(tab + nb * 16);
    d_isize = reinterpret_cast<uint32_t *>(tab + nb * 20);
    if (nb) {
        MTH_HIP(ctx, hipMemcpyAsync(d_coff, coff, nb * 8, hipMemcpyHostToDevice, s));
        MTH_HIP(ctx, hipMemcpyAsync(d_uoff, uoff.data(), nb * 8, hipMemcpyHostToDevice, s));
        MTH_HIP(ctx, hipMemcpyAsync(d_csize, csize, nb * 4, hipMemcpyHostToDevice, s));
        MTH_HIP(ctx, hipMemcpyAsync(d_isize, isize, nb * 4, hipMemcpyHostToDevice, s));
        MTH_HIP(ctx, hipStreamSynchronize(s));            // uoff is a local
        InflArgs ia{};
        ia.file = ctx->inf_file.as<uint8_t>(); ia.coff = d_coff; ia.csize = d_csize; ia.isize = d_isize; ia.uoff = d_uoff;
        ia.n_blocks = (uint32_t)nb; ia.out = ctx->inf_raw.as<uint8_t>(); ia.err = &ctx->d_state->err;
        { static const int fast = [] { const char *e = getenv("METHEOR_INFLATE_ASM"); return e ? atoi(e) : 1; }(); ia.fast_literals = fast; }
        auto launch = [&](size_t b0, size_t b1) {       // blocks [b0, b1): absolute output offsets, so a launch is any run of blocks
            InflArgs x = ia;
            x.coff = d_coff + b0; x.csize = d_csize + b0; x.isize = d_isize + b0; x.uoff = d_uoff + b0; x.n_blocks = (uint32_t)(b1 - b0);
            LaunchTimer lt(ctx, K_INFLATE);
            hipLaunchKernelGGL(k_inflate, dim3(x.n_blocks), dim3(64), 0, s, x);
        };
        if (pieced) {
            if (!ctx->piece_stream) MTH_HIP(ctx, hipStreamCreateWithFlags(&ctx->piece_stream, hipStreamNonBlocking));
            auto event = [&](size_t k) -> hipError_t {
                while (ctx->piece_ev.size() <= k) {
                    hipEvent_t e = nullptr;
                    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
                    if (rc != hipSuccess) return rc;
                    ctx->piece_ev.push_back(e);
                }
                return hipSuccess;
            };
            // whatever the compute stream still runs on the old contents of the buffer comes first
            MTH_HIP(ctx, event(0));
            MTH_HIP(ctx, hipEventRecord(ctx->piece_ev[0], s));
            MTH_HIP(ctx, hipStreamWaitEvent(ctx->piece_stream, ctx->piece_ev[0], 0));
            uint8_t *const dst = static_cast<uint8_t *>(ctx->inf_file.p);
            const uint8_t *const src = static_cast<const uint8_t *>(file);
            size_t off = 0, bl = 0, k = 1;
            while (off < (size_t)n_bytes) {
                size_t len = std::min(piece_bytes, (size_t)n_bytes - off);
                if ((size_t)n_bytes - off - len < piece_bytes / 2) len = (size_t)n_bytes - off;     // no stub piece at the end
                MTH_HIP(ctx, hipMemcpyAsync(dst + off, src + off, len, hipMemcpyHostToDevice, ctx->piece_stream));
                off += len;
                const bool last = off == (size_t)n_bytes;
                if (last) MTH_HIP(ctx, hipMemsetAsync(dst + n_bytes, 0, INF_FILE_PAD, ctx->piece_stream));
                MTH_HIP(ctx, event(k));
                MTH_HIP(ctx, hipEventRecord(ctx->piece_ev[k], ctx->piece_stream));
                MTH_HIP(ctx, hipStreamWaitEvent(s, ctx->piece_ev[k], 0));
                ++k;
                // the blocks whose payload -- and the 64 bytes the bit reader may load past it -- have arrived
                size_t b1 = bl;
                while (b1 < nb && (last || coff[b1] + (uint64_t)csize[b1] + 64u <= (uint64_t)off)) ++b1;
                if (b1 > bl) launch(bl, b1);
                bl = b1;
            }
        } else {
            launch(0, nb);
        }
        // CRC32 of every inflated block against the gzip trailer (the 4 bytes after the payload)
        for (size_t i = 0; i < nb; ++i)
            if (coff[i] + (uint64_t)csize[i] + 4 > n_bytes) return fail(ctx, MTH_ERR_INVALID, "BGZF block table: the CRC trailer of a block lies outside the file bytes");
        CrcArgs ca{};
        ca.raw = ctx->inf_raw.as<uint8_t>(); ca.file = ctx->inf_file.as<uint8_t>(); ca.coff = d_coff; ca.uoff = d_uoff;
        ca.csize = d_csize; ca.isize = d_isize; ca.mat = ctx->crc_mat.as<uint32_t>(); ca.n_blocks = (uint32_t)nb; ca.err = &ctx->d_state->err;
        LaunchTimer lt(ctx, K_CRC);
        hipLaunchKernelGGL(k_crc32, dim3((uint32_t)nb), dim3(64), 0, s, ca);
    }
    return MTH_OK;
}

extern "C" {

int mth_bgzf_stage(mth_ctx_t *ctx, const void *file, uint64_t n_bytes) {
    if (!ctx || (n_bytes && !file)) return MTH_ERR_INVALID;
    ctx->next_src = n_bytes ? file : nullptr;
    ctx->next_bytes = n_bytes;
    return MTH_OK;
}

int mth_bgzf_inflate(mth_ctx_t *ctx, const void *file, uint64_t n_bytes, const uint64_t *coff, const uint32_t *csize,
                     const uint32_t *isize, uint64_t n_blocks, void *dst_host, uint64_t *n_out) {
    if (!ctx || (n_blocks && (!file || !coff || !csize || !isize))) return MTH_ERR_INVALID;
    uint64_t total = 0, *d_uoff = nullptr;
    uint32_t *d_isize = nullptr;
    int rc = inflate_blocks(ctx, file, n_bytes, coff, csize, isize, n_blocks, total, d_uoff, d_isize);
    if (rc) return rc;
    if ((rc = sync_and_check(ctx))) return rc;
    if (n_out) *n_out = total;
    if (dst_host && total) MTH_HIP(ctx, hipMemcpy(dst_host, ctx->inf_raw.p, (size_t)total, hipMemcpyDeviceToHost));
    return MTH_OK;
}

int mth_bgzf_decode(mth_ctx_t *ctx, const void *file, uint64_t n_bytes, const uint64_t *coff, const uint32_t *csize,
                    const uint32_t *isize, uint64_t n_blocks, uint64_t first_byte, int append, mth_decoded_t *out) {
    if (!ctx || !out || (n_blocks && (!file || !coff || !csize || !isize))) return MTH_ERR_INVALID;
    uint64_t total = 0, *d_uoff = nullptr;
    uint32_t *d_isize = nullptr;
    int rc0 = inflate_blocks(ctx, file, n_bytes, coff, csize, isize, n_blocks, total, d_uoff, d_isize);
    if (rc0) return rc0;
    if (first_byte > total) return fail(ctx, MTH_ERR_INVALID, "first_byte beyond the inflated stream");
    hipStream_t s = ctx->stream;
    const size_t nb = (size_t)n_blocks;
    uint64_t n_rec = 0;
    if (nb) {
        // record offsets: count per block -> scan -> fill (verifying that no record straddles a block)
        MTH_HIP(ctx, ctx->inf_cnt.reserve(nb * 4 + 16, s));
        MTH_HIP(ctx, ctx->inf_base.reserve((nb + 1) * 8 + 16, s));
        WalkBArgs wa{};
        wa.raw = ctx->inf_raw.as<uint8_t>(); wa.uoff = d_uoff; wa.isize = d_isize; wa.n_blocks = (uint32_t)nb;
        wa.first_byte = first_byte; wa.cnt = ctx->inf_cnt.as<uint32_t>(); wa.total_bytes = total; wa.err = &ctx->d_state->err;
        hipLaunchKernelGGL((k_block_walk<false>), dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, wa);
        unsigned long long tot_rec = 0;
        int rc = scan_u32_to_u64(ctx, wa.cnt, (uint32_t)nb, 0ull, ctx->inf_base.as<unsigned long long>(), &tot_rec);
        if (rc) return rc;                       // (synchronises: inflate / walk errors surface here)
        n_rec = tot_rec;
        MTH_HIP(ctx, ctx->inf_recoff.reserve((size_t)(n_rec + 1) * 8 + 16, s));
        wa.base = ctx->inf_base.as<unsigned long long>(); wa.rec_off = ctx->inf_recoff.as<uint64_t>();
        hipLaunchKernelGGL((k_block_walk<true>), dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, wa);
    }
    return decode_core(ctx, ctx->inf_raw.as<uint8_t>(), ctx->inf_recoff.as<uint64_t>(), n_rec, append, out);
}

}  // extern "C"
