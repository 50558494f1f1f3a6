/*
 * brx_gzip.cpp -- multi-threaded gzip of the FASTQ stream for libbrx_host.so (include/brx_host.h).
 *
 * The input is cut into blocks of `block_bytes`; each block becomes one complete gzip member (deflateInit2 with
 * windowBits 15 + 16), compressed by a pool of std::threads into its own region of a scratch vector, and the
 * members are then laid end to end in input order.  Blocks do not share a dictionary, which costs ~1 % of ratio
 * at 1 MB blocks and buys linear scaling over cores.
 */
#include <zlib.h>

#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/brx_host.h"

static size_t member_bound(size_t n) {
    /* deflateBound for the default settings + gzip header/trailer; a little slack for level-0 stored blocks */
    return n + (n >> 12) + (n >> 14) + (n >> 25) + 13 + 18 + 64;
}

extern "C" size_t brx_gzip_bound(size_t n_bytes, size_t block_bytes) {
    if (block_bytes == 0) block_bytes = 1u << 20;
    const size_t blocks = n_bytes ? (n_bytes + block_bytes - 1) / block_bytes : 1;
    return blocks * member_bound(block_bytes < n_bytes ? block_bytes : n_bytes) + 64;
}

static int gzip_member(const uint8_t *in, size_t n, int level, uint8_t *out, size_t cap, size_t *written) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return BRX_E_INTERNAL;
    zs.next_in = const_cast<Bytef *>(in);
    zs.avail_in = (uInt)n;
    zs.next_out = out;
    zs.avail_out = (uInt)cap;
    const int rc = deflate(&zs, Z_FINISH);
    *written = (size_t)zs.total_out;
    deflateEnd(&zs);
    return rc == Z_STREAM_END ? BRX_OK : BRX_E_OUTPUT;
}

extern "C" int brx_gzip_parallel(const uint8_t *in, size_t n_bytes, int level, int threads, size_t block_bytes,
                                 uint8_t *out, size_t cap, size_t *out_bytes) {
    if ((!in && n_bytes) || !out || !out_bytes) return BRX_E_ARG;
    if (level < 0 || level > 9) level = 6;
    if (block_bytes == 0) block_bytes = 1u << 20;
    if (block_bytes > (1u << 30)) block_bytes = 1u << 30;            /* zlib's 32-bit avail_in */
    const size_t blocks = n_bytes ? (n_bytes + block_bytes - 1) / block_bytes : 1;   /* empty input: one empty member */
    if (threads < 1) threads = 1;
    if ((size_t)threads > blocks) threads = (int)blocks;
    const size_t bound = member_bound(block_bytes);
    std::vector<uint8_t> scratch(blocks * bound);
    std::vector<size_t> sizes(blocks, 0);
    std::atomic<size_t> next(0);
    std::atomic<int> status(BRX_OK);
    auto work = [&]() {
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= blocks) return;
            const size_t off = b * block_bytes;
            const size_t len = n_bytes > off ? ((n_bytes - off) < block_bytes ? (n_bytes - off) : block_bytes) : 0;
            const int rc = gzip_member(in + off, len, level, scratch.data() + b * bound, bound, &sizes[b]);
            if (rc != BRX_OK) status.store(rc);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
    if (status.load() != BRX_OK) return status.load();
    size_t total = 0;
    for (size_t b = 0; b < blocks; ++b) total += sizes[b];
    *out_bytes = total;
    if (total > cap) return BRX_E_OUTPUT;
    size_t at = 0;
    for (size_t b = 0; b < blocks; ++b) { memcpy(out + at, scratch.data() + b * bound, sizes[b]); at += sizes[b]; }
    return BRX_OK;
}
