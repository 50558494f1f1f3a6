/*
 * brx_gzip_dev.h -- the FASTQ bytes of a batch compressed on the GPU before they cross PCIe (SURVEY.md section 8f, row f2:
 * the output stage behind /root/reference/badread/simulate.py:73-82, where the reference prints text and leaves
 * compression to a `| gzip` pipe).
 *
 * Output: a sequence of independent gzip members (RFC 1952), one per BLOCK of input, which `gzip -d` and every gzip
 * reader treat as one stream.  Each member holds ONE deflate block (RFC 1951) with a dynamic Huffman code built from the
 * block's own byte histogram and NO match search: FASTQ from a simulator is near-random symbols -- bases with 2 bits of
 * entropy, qualities with ~5 -- which no short-window match search compresses (gzip -1 reaches 0.57 on this text) and an
 * order-0 coder is embarrassingly parallel.  The caller chooses the blocks.  With none given they are 64 KB each and one
 * code serves bases and qualities alike (~0.56, gzip -1's ratio).  The driver cuts at the LINES instead (header + sequence
 * line | '+' + quality line, short reads merged: badread_amd/output.py): a code per line type brings the text to ~0.45.
 *
 *   k_gz_plan   one wave per block: byte histogram (LDS atomics), length-limited Huffman code (rank sort across
 *               the lanes, two-queue merge and Kraft repair on one lane: < 80 symbols occur), canonical codes; then every
 *               lane walks its own chunk (1/64 of the block) once for its bit count and its CRC-32; the 64 chunk CRCs are
 *               folded with the x^n-mod-P operator (the algebra zlib's crc32_combine uses: crc(A||B) = crc(A) * x^(8|B|) + crc(B))
 *   k_gz_scan   exclusive scan of the member sizes (one wave)
 *   k_gz_pack   one wave per block: gzip header, the 1110 header bits of the dynamic block (all 19 code-length codes
 *               sent, code lengths 0..15 as fixed 4-bit codes: no run-length symbols to search for), every lane packs its
 *               chunk LSB-first into a 64-bit accumulator and emits 32-bit words -- plain stores for the words it owns, an
 *               atomic OR for the two it shares with its neighbours -- end-of-block code, CRC-32 and length
 *
 * Memory-bound byte work (three reads of the input, one write of ~0.42 of it); no MFMA, no LDS tiling beyond the tables.
 */
#ifndef BRX_GZIP_DEV_H
#define BRX_GZIP_DEV_H

#define BRX_GZ_BLOCK 65536u                  /* input bytes per gzip member when the caller gives no blocks */
#define BRX_GZ_MAX_BLOCK (1u << 27)          /* longest block (bit offsets are 32-bit)               */
#define BRX_GZ_SYMS 257u                     /* literals 0..255 + end of block                       */
#define BRX_GZ_HDR_BITS 1110u                /* 3 + 5 + 5 + 4 + 19 x 3 + (257 + 2) x 4               */
#define BRX_GZ_TAB 260u                      /* table words stored per block (257 used)              */
#define BRX_GZ_OFFS 65u                      /* per block: 64 lane bit offsets + the total           */
#define BRX_GZ_POLY 0xEDB88320u

struct BrxGzConst { uint32_t x2n[32]; };     /* x^(2^k) mod P, reflected: x2n[0] = 0x40000000 */

/* member bytes a block of `len` input bytes can need at most (15-bit codes) */
__host__ __device__ inline uint64_t brx_gz_member_bound(uint64_t len) { return 10 + (BRX_GZ_HDR_BITS + 15ull * (len + 1) + 7) / 8 + 8; }

/* (a * b) mod P over GF(2), reflected representation (bit 31 = x^0) */
__host__ __device__ inline uint32_t brx_gz_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ BRX_GZ_POLY : b >> 1;
    }
    return p;
}
/* x^(n * 2^k) mod P */
__host__ __device__ inline uint32_t brx_gz_x2nmodp(const uint32_t *x2n, uint64_t n, unsigned k) {
    uint32_t p = 0x80000000u;
    while (n) {
        if (n & 1u) p = brx_gz_mulmod(x2n[k & 31u], p);
        n >>= 1; k++;
    }
    return p;
}

__device__ __forceinline__ uint32_t brx_gz_rev(uint32_t code, uint32_t len) {      /* the low `len` bits, reversed */
    uint32_t r = 0;
    for (uint32_t i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1u - i);
    return r;
}

/* block b covers input bytes [blk_off[b], blk_off[b + 1]) -- or 64 KB pieces of [0, n) when blk_off is null */
__device__ __forceinline__ void brx_gz_block(const uint64_t *blk_off, uint64_t n, uint32_t b, uint64_t *base, uint32_t *len) {
    if (blk_off) { *base = blk_off[b]; *len = (uint32_t)(blk_off[b + 1] - blk_off[b]); }
    else { *base = (uint64_t)b * BRX_GZ_BLOCK; *len = (uint32_t)(n - *base < BRX_GZ_BLOCK ? n - *base : BRX_GZ_BLOCK); }
}

__global__ void __launch_bounds__(64) k_gz_plan(const uint8_t *__restrict__ in, uint64_t n, const uint64_t *__restrict__ blk_off, uint32_t n_blocks, BrxGzConst K,
                                                uint32_t *__restrict__ tabs, uint32_t *__restrict__ offs, uint32_t *__restrict__ crcs,
                                                uint32_t *__restrict__ sizes) {
    __shared__ uint32_t hist[BRX_GZ_TAB], tab[BRX_GZ_TAB], crc_tab[256];
    __shared__ uint32_t weight[2 * BRX_GZ_TAB];
    __shared__ uint16_t parent[2 * BRX_GZ_TAB], ord[BRX_GZ_TAB];
    __shared__ uint8_t depth[2 * BRX_GZ_TAB], lens[BRX_GZ_TAB];
    const int lane = threadIdx.x & 63;
    for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        uint64_t base; uint32_t len;
        brx_gz_block(blk_off, n, b, &base, &len);
        const uint32_t chunk = (len + 63u) / 64u;                  /* bytes per lane */
        for (uint32_t s = lane; s < BRX_GZ_TAB; s += 64) { hist[s] = 0; lens[s] = 0; tab[s] = 0; }
        for (uint32_t s = lane; s < 256; s += 64) {
            uint32_t c = s;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ BRX_GZ_POLY : c >> 1;
            crc_tab[s] = c;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        /* ---- histogram: 64 bytes per step, coalesced (blocks start at any byte) ---- */
        for (uint32_t i = (uint32_t)lane; i < len; i += 64u) atomicAdd(&hist[in[base + i]], 1u);
        if (lane == 0) hist[256] = 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        /* ---- symbols in use, ascending by (count, symbol): rank sort, five symbols per lane ---- */
        uint32_t n_act = 0;
        for (uint32_t s0 = 0; s0 < BRX_GZ_SYMS; s0 += 64) {
            const uint32_t s = s0 + (uint32_t)lane;
            n_act += (uint32_t)__popcll(__ballot(s < BRX_GZ_SYMS && hist[s] > 0));
        }
        for (uint32_t s = lane; s < BRX_GZ_SYMS; s += 64) {
            const uint32_t cs = hist[s];
            if (!cs) continue;
            uint32_t rank = 0;
            for (uint32_t t = 0; t < BRX_GZ_SYMS; ++t) { const uint32_t ct = hist[t]; rank += (ct > 0 && (ct < cs || (ct == cs && t < s))) ? 1u : 0u; }
            ord[rank] = (uint16_t)s;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) {
            /* ---- Huffman: two queues (sorted leaves, internal nodes in creation order) ---- */
            for (uint32_t i = 0; i < n_act; ++i) weight[i] = hist[ord[i]];
            uint32_t li = 0, ii = n_act, ni = n_act;
            auto pick = [&]() -> uint32_t {
                if (li < n_act && (ii >= ni || weight[li] <= weight[ii])) return li++;
                return ii++;
            };
            for (uint32_t m = 1; m < n_act; ++m) {
                const uint32_t a = pick(), c2 = pick();
                weight[ni] = weight[a] + weight[c2];
                parent[a] = (uint16_t)ni; parent[c2] = (uint16_t)ni;
                ni += 1;
            }
            const uint32_t root = ni - 1;
            depth[root] = 0;
            for (int i = (int)root - 1; i >= 0; --i) depth[i] = (uint8_t)(depth[parent[i]] + 1);
            if (n_act == 1) depth[0] = 1;
            /* ---- at most 15 bits: clamp, then repair the Kraft sum (units of 2^-15) ---- */
            uint32_t kraft = 0;
            for (uint32_t i = 0; i < n_act; ++i) { uint32_t L = depth[i]; if (L > 15) L = 15; lens[ord[i]] = (uint8_t)L; kraft += 1u << (15 - L); }
            while (kraft > 32768u) {                       /* lengthen the rarest symbol that is not at 15 yet */
                for (uint32_t i = 0; i < n_act; ++i) {
                    const uint32_t L = lens[ord[i]];
                    if (L < 15) { lens[ord[i]] = (uint8_t)(L + 1); kraft -= 1u << (14 - L); break; }
                }
            }
            while (kraft < 32768u) {                       /* shorten the most frequent symbol that fits the slack */
                bool moved = false;
                for (int i = (int)n_act - 1; i >= 0; --i) {
                    const uint32_t L = lens[ord[i]];
                    if (L > 1 && (1u << (15 - L)) <= 32768u - kraft) { lens[ord[i]] = (uint8_t)(L - 1); kraft += 1u << (15 - L); moved = true; break; }
                }
                if (!moved) break;
            }
            /* ---- canonical codes (RFC 1951 3.2.2), stored bit-reversed: the stream is filled from the low bit ---- */
            uint32_t bl_count[16], next_code[16];
            for (int L = 0; L < 16; ++L) bl_count[L] = 0;
            for (uint32_t s = 0; s < BRX_GZ_SYMS; ++s) bl_count[lens[s]] += 1;
            bl_count[0] = 0;
            uint32_t code = 0;
            for (int L = 1; L < 16; ++L) { code = (code + bl_count[L - 1]) << 1; next_code[L] = code; }
            for (uint32_t s = 0; s < BRX_GZ_SYMS; ++s) {
                const uint32_t L = lens[s];
                if (L) { tab[s] = brx_gz_rev(next_code[L], L) | (L << 16); next_code[L] += 1; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        /* ---- every lane walks its chunk: bits of its codes, CRC-32 of its bytes ---- */
        const uint32_t c_begin = chunk * (uint32_t)lane;
        const uint32_t c_len = c_begin < len ? (len - c_begin < chunk ? len - c_begin : chunk) : 0u;
        uint32_t bits = 0, crc = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < c_len; ++i) {
            const uint32_t ch = in[base + c_begin + i];
            bits += tab[ch] >> 16;
            crc = crc_tab[(crc ^ ch) & 0xFFu] ^ (crc >> 8);
        }
        crc ^= 0xFFFFFFFFu;
        uint32_t incl = bits;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, (unsigned)d, 64); if (lane >= d) incl += o; }
        const uint32_t total_bits = (uint32_t)__shfl((int)incl, 63, 64);
        offs[(uint64_t)b * BRX_GZ_OFFS + (uint32_t)lane] = incl - bits;
        /* ---- fold the chunk CRCs: crc(A || B) = crc(A) * x^(8 |B|) + crc(B) ---- */
        uint32_t sub_len = c_len;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const uint32_t r_crc = (uint32_t)__shfl_down((int)crc, 1u << k, 64), r_len = (uint32_t)__shfl_down((int)sub_len, 1u << k, 64);
            if ((lane & ((2 << k) - 1)) == 0) {
                const uint32_t op = (chunk == 1024u && r_len == (1024u << k)) ? K.x2n[(13 + k) & 31] : brx_gz_x2nmodp(K.x2n, r_len, 3);
                crc = brx_gz_mulmod(op, crc) ^ r_crc;
                sub_len += r_len;
            }
        }
        for (uint32_t s = lane; s < BRX_GZ_TAB; s += 64) tabs[(uint64_t)b * BRX_GZ_TAB + s] = tab[s];
        if (lane == 0) {
            offs[(uint64_t)b * BRX_GZ_OFFS + 64] = total_bits;
            crcs[b] = crc;
            sizes[b] = 10u + (BRX_GZ_HDR_BITS + total_bits + (tab[256] >> 16) + 7u) / 8u + 8u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
    }
}

/* member_off[b] = sum of sizes[0..b), member_off[n_blocks] = total: one wave */
__global__ void __launch_bounds__(64) k_gz_scan(uint32_t n_blocks, const uint32_t *__restrict__ sizes, uint64_t *__restrict__ member_off) {
    const int lane = threadIdx.x & 63;
    uint64_t running = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 64) {
        const uint32_t b = b0 + (uint32_t)lane;
        const uint32_t v = b < n_blocks ? sizes[b] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, (unsigned)d, 64); if (lane >= d) incl += o; }
        if (b < n_blocks) member_off[b] = running + incl - v;
        running += (uint64_t)(uint32_t)__shfl((int)incl, 63, 64);
    }
    if (lane == 0) member_off[n_blocks] = running;
}

/* `value` (nbits <= 25 low bits) at absolute bit position `pos` of the zero-initialised output */
__device__ __forceinline__ void brx_gz_put(uint32_t *out32, uint64_t pos, uint32_t value, uint32_t nbits) {
    const uint64_t w = pos >> 5;
    const uint32_t sh = (uint32_t)(pos & 31u);
    atomicOr(&out32[w], value << sh);
    if (sh + nbits > 32u) atomicOr(&out32[w + 1], value >> (32u - sh));
}

__global__ void __launch_bounds__(64) k_gz_pack(const uint8_t *__restrict__ in, uint64_t n, const uint64_t *__restrict__ blk_off, uint32_t n_blocks, uint8_t *__restrict__ out,
                                                const uint64_t *__restrict__ member_off, const uint32_t *__restrict__ tabs,
                                                const uint32_t *__restrict__ offs, const uint32_t *__restrict__ crcs) {
    __shared__ uint32_t tab[BRX_GZ_TAB];
    const int lane = threadIdx.x & 63;
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
    for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        uint64_t base; uint32_t len;
        brx_gz_block(blk_off, n, b, &base, &len);
        const uint32_t chunk = (len + 63u) / 64u;
        const uint64_t mo = member_off[b], mend = member_off[b + 1];
        for (uint32_t s = lane; s < BRX_GZ_TAB; s += 64) tab[s] = tabs[(uint64_t)b * BRX_GZ_TAB + s];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t bit0 = 8ull * (mo + 10ull);                     /* first bit of the deflate stream */
        if (lane == 0) {
            const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
            for (int i = 0; i < 10; ++i) out[mo + (uint64_t)i] = hdr[i];
            brx_gz_put(out32, bit0, 1u | (2u << 1), 3);                /* BFINAL = 1, BTYPE = 2 (dynamic Huffman codes) */
            brx_gz_put(out32, bit0 + 3, 0u | (1u << 5) | (15u << 10), 14);     /* HLIT = 257 - 257, HDIST = 2 - 1, HCLEN = 19 - 4 */
            /* code lengths of the code-length alphabet, in the order 16 17 18 0 8 7 ...: 0 for the run-length symbols,
               4 for the lengths 0..15, whose canonical codes are then the 4-bit numbers themselves */
            for (uint32_t i = 0; i < 19; ++i) brx_gz_put(out32, bit0 + 17 + 3ull * i, i < 3 ? 0u : 4u, 3);
            const uint32_t crc = crcs[b];
            for (int i = 0; i < 4; ++i) { out[mend - 8 + (uint64_t)i] = (uint8_t)(crc >> (8 * i)); out[mend - 4 + (uint64_t)i] = (uint8_t)(len >> (8 * i)); }
        }
        for (uint32_t s = lane; s < BRX_GZ_SYMS + 2u; s += 64) {      /* 257 literal/length code lengths, 2 distance code lengths of 1 */
            const uint32_t L = s < BRX_GZ_SYMS ? tab[s] >> 16 : 1u;
            brx_gz_put(out32, bit0 + 74 + 4ull * s, ((L & 1u) << 3) | ((L & 2u) << 1) | ((L & 4u) >> 1) | ((L & 8u) >> 3), 4);
        }
        /* ---- the lane's chunk, LSB first ---- */
        const uint32_t c_begin = chunk * (uint32_t)lane;
        const uint32_t c_len = c_begin < len ? (len - c_begin < chunk ? len - c_begin : chunk) : 0u;
        const uint64_t data0 = bit0 + BRX_GZ_HDR_BITS;
        uint64_t pos = data0 + offs[(uint64_t)b * BRX_GZ_OFFS + (uint32_t)lane];
        uint64_t widx = pos >> 5;
        uint32_t accbits = (uint32_t)(pos & 31u);                       /* the low bits of the first word belong to a neighbour */
        uint64_t acc = 0;
        bool first = true;
        auto append = [&](uint32_t e) {
            acc |= (uint64_t)(e & 0xFFFFu) << accbits;
            accbits += e >> 16;
            if (accbits >= 32u) {
                if (first) atomicOr(&out32[widx], (uint32_t)acc); else out32[widx] = (uint32_t)acc;
                first = false;
                acc >>= 32; accbits -= 32u; widx += 1;
            }
        };
        for (uint32_t i = 0; i < c_len; ++i) append(tab[in[base + c_begin + i]]);
        if (accbits > 0 && acc != 0) atomicOr(&out32[widx], (uint32_t)acc);
        if (lane == 0) brx_gz_put(out32, data0 + offs[(uint64_t)b * BRX_GZ_OFFS + 64], tab[256] & 0xFFFFu, tab[256] >> 16);     /* end of block */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
    }
}

#endif /* BRX_GZIP_DEV_H */
