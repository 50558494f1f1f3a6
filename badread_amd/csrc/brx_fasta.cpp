/*
 * brx_fasta.cpp -- FASTA(.gz) -> 2-bit packed reference for libbrx_host.so (include/brx_host.h).
 *
 * One streaming pass: zlib's gz* layer reads plain and gzip files alike; each line is stripped and classified
 * the way misc.load_fasta does (/root/reference/badread/misc.py:122-153); bases go straight into the packed
 * words, everything outside ACGT into a run list keyed by the raw symbol, and the symbol codes (which depend
 * on every symbol of the file) are resolved at the end.  Memory: the packed genome + O(runs).
 */
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <unordered_map>
#include <vector>

#include "../../include/brx_host.h"

namespace {

struct Run { uint64_t start, end; uint8_t symbol; };

struct Contig {
    std::string name;
    double depth = 1.0;
    uint32_t flags = 0;
    uint64_t length = 0;
    std::vector<uint32_t> words;      /* own packing, base 0 at bit 0: contigs are spliced together at the end */
    std::vector<Run> runs;            /* contig coordinates */
    uint32_t cur = 0;                 /* partial word */
};

int set_err(char *err, size_t cap, int code, const char *fmt, ...) {
    if (err && cap) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, cap, fmt, ap);
        va_end(ap);
    }
    return code;
}

inline bool is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

uint8_t complement_of(uint8_t c) {            /* misc.py:56-67 on upper-cased symbols; unknown -> N */
    switch (c) {
        case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G';
        case 'R': return 'Y'; case 'Y': return 'R'; case 'S': return 'S'; case 'W': return 'W';
        case 'K': return 'M'; case 'M': return 'K'; case 'B': return 'V'; case 'V': return 'B';
        case 'D': return 'H'; case 'H': return 'D'; case 'N': return 'N';
        case '.': return '.'; case '-': return '-'; case '?': return '?';
        /* lower-case keys of the table can only be reached by symbols str.upper() leaves alone: none in ASCII */
        default: return 'N';
    }
}

/* depth=([\d.]+) searched in the lower-cased header; float() must accept the match, else 1.0 */
double parse_depth(const std::string &lowered) {
    size_t at = lowered.find("depth=");
    while (at != std::string::npos) {
        size_t b = at + 6, e = b;
        while (e < lowered.size() && (std::isdigit((unsigned char)lowered[e]) || lowered[e] == '.')) ++e;
        if (e > b) {
            const std::string tok = lowered.substr(b, e - b);
            int dots = 0;
            for (char ch : tok) dots += ch == '.';
            if (dots > 1 || tok == ".") return 1.0;           /* float('1.2.3') raises -> 1.0 */
            return strtod(tok.c_str(), nullptr);
        }
        at = lowered.find("depth=", at + 1);                   /* regex search moves on to the next 'depth=' */
    }
    return 1.0;
}

}  // namespace

struct brx_fasta {
    std::vector<uint32_t> packed;
    std::vector<brx_contig> contigs;
    std::vector<brx_exception> exceptions;
    std::vector<uint8_t> names;
    std::vector<double> depths;
    uint64_t n_bases = 0;
    uint32_t n_symbols = 5;
    uint8_t sym[16], comp[16];
    /* a reference loaded from its sidecar keeps the packed words IN the mapped file (772 MB for a human genome: no read into a
       zero-filled vector, no second copy): `packed` stays empty and the view points into the mapping */
    const uint32_t *mapped_packed = nullptr;
    size_t mapped_words = 0;
    void *map_base = nullptr;
    size_t map_len = 0;
    ~brx_fasta() { if (map_base) munmap(map_base, map_len); }
};

/* byte -> 2-bit code of A,C,G,T in either case; 0xFF for everything else */
static const uint8_t *base_lut() {
    static uint8_t lut[256];
    static bool ready = false;
    if (!ready) {
        memset(lut, 0xFF, sizeof(lut));
        lut['A'] = lut['a'] = 0; lut['C'] = lut['c'] = 1; lut['G'] = lut['g'] = 2; lut['T'] = lut['t'] = 3;
        ready = true;
    }
    return lut;
}

static void append_bases(Contig &c, const unsigned char *p, size_t n) {
    const uint8_t *lut = base_lut();
    uint64_t len = c.length;
    uint32_t cur = c.cur;
    for (size_t i = 0; i < n; ++i) {
        uint32_t code = lut[p[i]];
        if (__builtin_expect(code == 0xFFu, 0)) {
            unsigned char ch = p[i];
            if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 32);          /* str.upper() */
            code = 0;
            if (!c.runs.empty() && c.runs.back().end == len && c.runs.back().symbol == ch) c.runs.back().end += 1;
            else c.runs.push_back({len, len + 1, ch});
        }
        cur |= code << (2u * (uint32_t)(len & 15u));
        len += 1;
        if ((len & 15u) == 0) { c.words.push_back(cur); cur = 0; }
    }
    c.length = len;
    c.cur = cur;
}

extern "C" int brx_fasta_pack(const char *path, brx_fasta **out, char *err, size_t err_cap) {
    if (!path || !out) return set_err(err, err_cap, BRX_E_ARG, "brx_fasta_pack: null argument");
    *out = nullptr;
    gzFile fp = gzopen(path, "rb");
    if (!fp) return set_err(err, err_cap, BRX_E_ARG, "could not open %s", path);
    gzbuffer(fp, 1 << 20);

    std::vector<Contig> contigs;
    std::unordered_map<std::string, size_t> index;
    long cur = -1;                      /* contig receiving sequence lines; -1 before the first header */
    std::vector<unsigned char> buf(1 << 20);
    bool any = false;

    /* Byte-level line machine, so that a whole chromosome on one line is never buffered:
     *   START   only blanks seen on this line        HEADER  collecting the text after '>'
     *   SEQ     streaming bases into `cur`; blanks are held back until a non-blank follows (str.strip() drops
     *           them at the end of the line, keeps them inside it)                                            */
    enum { START, HEADER, SEQ } mode = START;
    std::string header;
    std::string held;                   /* blanks inside a sequence line, not yet known to be interior */
    std::vector<unsigned char> orphan;  /* sequence lines before the first header: the reference's loop keeps them in its
                                           line list and they become the start of the FIRST contig (misc.py:131-134) */

    auto end_header = [&]() -> int {
        size_t n = header.size();
        while (n && is_space((unsigned char)header[n - 1])) --n;
        header.resize(n);
        size_t b = 0;
        while (b < header.size() && is_space((unsigned char)header[b])) ++b;
        size_t e = b;
        while (e < header.size() && !is_space((unsigned char)header[e])) ++e;
        if (e == b) return -1;                                      /* the reference raises IndexError here */
        const std::string name = header.substr(b, e - b);
        std::string lowered = header;
        for (char &ch : lowered) if (ch >= 'A' && ch <= 'Z') ch = (char)(ch + 32);
        Contig fresh;
        fresh.name = name;
        fresh.depth = parse_depth(lowered);
        fresh.flags = (lowered.find("circular=true") != std::string::npos ? 1u : 0u) |
                      (lowered.find("hairpin_left=true") != std::string::npos ? 2u : 0u) |
                      (lowered.find("hairpin_right=true") != std::string::npos ? 4u : 0u);
        auto it = index.find(name);
        if (contigs.empty() && !orphan.empty()) { append_bases(fresh, orphan.data(), orphan.size()); orphan.clear(); }
        if (it == index.end()) { index[name] = contigs.size(); contigs.push_back(std::move(fresh)); cur = (long)contigs.size() - 1; }
        else { contigs[it->second] = std::move(fresh); cur = (long)it->second; }     /* dict assignment: first position, last value */
        return 0;
    };

    for (;;) {
        const int got = gzread(fp, buf.data(), (unsigned)buf.size());
        if (got < 0) { gzclose(fp); return set_err(err, err_cap, BRX_E_ARG, "read error in %s", path); }
        if (got == 0) break;
        any = true;
        size_t i = 0;
        const size_t n = (size_t)got;
        while (i < n) {
            const unsigned char ch = buf[i];
            if (ch == '\n' || ch == '\r') {                           /* text mode: both end a line */
                if (mode == HEADER && end_header()) { gzclose(fp); return set_err(err, err_cap, BRX_E_ARG, "empty FASTA header in %s", path); }
                mode = START; header.clear(); held.clear();
                ++i;
                continue;
            }
            if (mode == START) {
                if (is_space(ch)) { ++i; continue; }
                if (ch == '>') { mode = HEADER; ++i; continue; }
                mode = SEQ;
            }
            if (mode == HEADER) {
                size_t e = i;
                while (e < n && buf[e] != '\n' && buf[e] != '\r') ++e;
                header.append((const char *)buf.data() + i, e - i);
                i = e;
                continue;
            }
            /* SEQ */
            if (is_space(ch)) { held.push_back((char)ch); ++i; continue; }
            size_t e = i;
            while (e < n && !is_space(buf[e])) ++e;                  /* '\n' and '\r' are blanks too */
            if (cur >= 0) {
                if (!held.empty()) append_bases(contigs[(size_t)cur], (const unsigned char *)held.data(), held.size());
                append_bases(contigs[(size_t)cur], buf.data() + i, e - i);
            } else {
                orphan.insert(orphan.end(), held.begin(), held.end());
                orphan.insert(orphan.end(), buf.data() + i, buf.data() + e);
            }
            held.clear();
            i = e;
        }
    }
    gzclose(fp);
    if (mode == HEADER && end_header()) return set_err(err, err_cap, BRX_E_ARG, "empty FASTA header in %s", path);
    if (!any || contigs.empty()) return set_err(err, err_cap, BRX_E_ARG, "%s holds no FASTA records", path);

    /* ---- alphabet ---- */
    bool present[256] = {false};
    for (const Contig &c : contigs) for (const Run &r : c.runs) present[r.symbol] = true;
    std::vector<uint8_t> symbols = {'A', 'C', 'G', 'T', 'N'};
    auto has = [&](uint8_t s) { return std::find(symbols.begin(), symbols.end(), s) != symbols.end(); };
    for (int b = 0; b < 256; ++b) {
        if (!present[b] || has((uint8_t)b)) continue;
        symbols.push_back((uint8_t)b);
        const uint8_t cb = complement_of((uint8_t)b);
        if (!has(cb)) symbols.push_back(cb);
    }
    if (symbols.size() > 16) return set_err(err, err_cap, BRX_E_ARG, "reference uses more than 16 distinct symbols");
    brx_fasta *f = new brx_fasta();
    uint8_t code_of[256];
    memset(code_of, 4, sizeof(code_of));
    for (size_t i = 0; i < symbols.size(); ++i) { f->sym[i] = symbols[i]; code_of[symbols[i]] = (uint8_t)i; }
    for (size_t i = 0; i < symbols.size(); ++i) f->comp[i] = code_of[complement_of(symbols[i])];
    for (size_t i = symbols.size(); i < 16; ++i) { f->sym[i] = 'N'; f->comp[i] = 4; }
    f->n_symbols = (uint32_t)symbols.size();

    /* ---- splice the contigs into one base-indexed array ---- */
    uint64_t total = 0;
    for (Contig &c : contigs) {
        if (c.length >= (1ull << 32)) { delete f; return set_err(err, err_cap, BRX_E_ARG, "contig %s is longer than 2^32-1 bases", c.name.c_str()); }
        if (c.length & 15u) c.words.push_back(c.cur);
        total += c.length;
    }
    f->n_bases = total;
    f->packed.assign((size_t)((total + 15) / 16 + 1), 0u);
    uint64_t g = 0;
    for (const Contig &c : contigs) {
        brx_contig d;
        d.base_off = g; d.length = (uint32_t)c.length; d.flags = c.flags;
        d.name_off = (uint32_t)f->names.size(); d.name_len = (uint32_t)c.name.size();
        f->names.insert(f->names.end(), c.name.begin(), c.name.end());
        f->contigs.push_back(d);
        f->depths.push_back(c.depth);
        const uint32_t sh = 2u * (uint32_t)(g & 15u);
        uint64_t w = g >> 4;
        const size_t nw = c.words.size();
        if (sh == 0) {
            if (nw) memcpy(&f->packed[(size_t)w], c.words.data(), nw * 4);
        } else {
            for (size_t i = 0; i < nw; ++i) {
                const uint32_t v = c.words[i];
                f->packed[(size_t)(w + i)] |= v << sh;
                f->packed[(size_t)(w + i + 1)] |= v >> (32u - sh);
            }
        }
        for (const Run &r : c.runs) {
            const uint32_t code = code_of[r.symbol];
            if (!f->exceptions.empty() && f->exceptions.back().end == g + r.start && f->exceptions.back().code == code)
                f->exceptions.back().end = g + r.end;                 /* runs merge across contig boundaries, as in reference.py */
            else f->exceptions.push_back(brx_exception{g + r.start, g + r.end, code, 0u});
        }
        g += c.length;
    }
    /* the unused high bits of the last word pair stay zero because every contig's tail word is zero-padded */
    *out = f;
    return BRX_OK;
}

extern "C" int brx_fasta_view_of(const brx_fasta *f, brx_fasta_view *v) {
    if (!f || !v) return BRX_E_ARG;
    v->n_bases = f->n_bases; v->n_words = f->mapped_packed ? f->mapped_words : f->packed.size();
    v->n_contigs = (uint32_t)f->contigs.size(); v->n_exceptions = (uint32_t)f->exceptions.size();
    v->names_len = (uint32_t)f->names.size(); v->n_symbols = f->n_symbols;
    v->packed = f->mapped_packed ? f->mapped_packed : f->packed.data(); v->contigs = f->contigs.data();
    v->exceptions = f->exceptions.empty() ? nullptr : f->exceptions.data();
    v->names = f->names.data(); v->depths = f->depths.data();
    memcpy(v->sym, f->sym, 16); memcpy(v->comp, f->comp, 16);
    return BRX_OK;
}

extern "C" void brx_fasta_free(brx_fasta *f) { delete f; }

/* ---- sidecar ------------------------------------------------------------------------------------------ */
namespace {
struct SidecarHeader {
    char magic[8];                 /* "BRX2BIT\2": the packed words start at the next multiple of 8 bytes behind the names */
    uint64_t src_size; int64_t src_mtime_ns;
    uint64_t n_bases, n_words;
    uint32_t n_contigs, n_exceptions, names_len, n_symbols;
    uint8_t sym[16], comp[16];
};
const char MAGIC[8] = {'B', 'R', 'X', '2', 'B', 'I', 'T', 2};

bool stat_source(const char *path, uint64_t *size, int64_t *mtime_ns) {
    struct stat st;
    if (stat(path, &st) != 0) return false;
    *size = (uint64_t)st.st_size;
    *mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
    return true;
}
}  // namespace

extern "C" int brx_fasta_save(const brx_fasta *f, const char *source_path, const char *sidecar_path, char *err, size_t err_cap) {
    if (!f || !source_path || !sidecar_path) return set_err(err, err_cap, BRX_E_ARG, "brx_fasta_save: null argument");
    SidecarHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, MAGIC, 8);
    if (!stat_source(source_path, &h.src_size, &h.src_mtime_ns)) return set_err(err, err_cap, BRX_E_ARG, "could not stat %s", source_path);
    const size_t n_words = f->mapped_packed ? f->mapped_words : f->packed.size();
    const uint32_t *words = f->mapped_packed ? f->mapped_packed : f->packed.data();
    h.n_bases = f->n_bases; h.n_words = n_words;
    h.n_contigs = (uint32_t)f->contigs.size(); h.n_exceptions = (uint32_t)f->exceptions.size();
    h.names_len = (uint32_t)f->names.size(); h.n_symbols = f->n_symbols;
    memcpy(h.sym, f->sym, 16); memcpy(h.comp, f->comp, 16);
    /* a name of this process's own next to the target (mkstemp): the ranks of a first multi-GPU run pack and save at the same
       time, and a shared "<sidecar>.tmp" let one rank truncate what another was writing or had just renamed into place */
    std::string tmp = std::string(sidecar_path) + ".XXXXXX";
    const int fd = mkstemp(&tmp[0]);
    FILE *fp = fd >= 0 ? fdopen(fd, "wb") : nullptr;
    if (!fp) { if (fd >= 0) { close(fd); remove(tmp.c_str()); } return set_err(err, err_cap, BRX_E_ARG, "could not write %s", tmp.c_str()); }
    (void)fchmod(fd, 0644);                                     /* mkstemp creates 0600: the cache is as readable as fopen would have made it */
    bool ok = fwrite(&h, sizeof(h), 1, fp) == 1;
    ok = ok && fwrite(f->contigs.data(), sizeof(brx_contig), f->contigs.size(), fp) == f->contigs.size();
    ok = ok && fwrite(f->depths.data(), sizeof(double), f->depths.size(), fp) == f->depths.size();
    ok = ok && fwrite(f->exceptions.data(), sizeof(brx_exception), f->exceptions.size(), fp) == f->exceptions.size();
    ok = ok && fwrite(f->names.data(), 1, f->names.size(), fp) == f->names.size();
    {   /* the words are read in place from the mapped file: aligned */
        const size_t at = sizeof(h) + f->contigs.size() * sizeof(brx_contig) + f->depths.size() * sizeof(double) +
                          f->exceptions.size() * sizeof(brx_exception) + f->names.size();
        const char zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const size_t pad = (8 - at % 8) % 8;
        ok = ok && (pad == 0 || fwrite(zeros, 1, pad, fp) == pad);
    }
    ok = ok && fwrite(words, 4, n_words, fp) == n_words;
    ok = (fclose(fp) == 0) && ok;
    if (!ok || rename(tmp.c_str(), sidecar_path) != 0) { remove(tmp.c_str()); return set_err(err, err_cap, BRX_E_ARG, "could not write %s", sidecar_path); }
    return BRX_OK;
}

extern "C" int brx_fasta_load(const char *source_path, const char *sidecar_path, brx_fasta **out, char *err, size_t err_cap) {
    if (!source_path || !sidecar_path || !out) return set_err(err, err_cap, BRX_E_ARG, "brx_fasta_load: null argument");
    *out = nullptr;
    uint64_t size; int64_t mtime;
    if (!stat_source(source_path, &size, &mtime)) return set_err(err, err_cap, BRX_E_STATE, "could not stat %s", source_path);
    const int fd = open(sidecar_path, O_RDONLY);
    if (fd < 0) return set_err(err, err_cap, BRX_E_STATE, "no sidecar %s", sidecar_path);
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(SidecarHeader)) { close(fd); return set_err(err, err_cap, BRX_E_STATE, "sidecar %s is truncated", sidecar_path); }
    const size_t len = (size_t)st.st_size;
    void *base = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (base == MAP_FAILED) return set_err(err, err_cap, BRX_E_STATE, "could not map %s", sidecar_path);
    SidecarHeader h;
    memcpy(&h, base, sizeof(h));
    if (memcmp(h.magic, MAGIC, 8) != 0 || h.src_size != size || h.src_mtime_ns != mtime ||
        h.n_words != (h.n_bases + 15) / 16 + 1 || h.n_symbols > 16) {
        munmap(base, len);
        return set_err(err, err_cap, BRX_E_STATE, "sidecar %s does not match %s", sidecar_path, source_path);
    }
    size_t at = sizeof(h);
    const size_t contigs_at = at; at += (size_t)h.n_contigs * sizeof(brx_contig);
    const size_t depths_at = at; at += (size_t)h.n_contigs * sizeof(double);
    const size_t exc_at = at; at += (size_t)h.n_exceptions * sizeof(brx_exception);
    const size_t names_at = at; at += (size_t)h.names_len;
    at = (at + 7) & ~(size_t)7;
    if (at + (size_t)h.n_words * 4 > len) { munmap(base, len); return set_err(err, err_cap, BRX_E_STATE, "sidecar %s is truncated", sidecar_path); }
    brx_fasta *f = new brx_fasta();
    f->n_bases = h.n_bases; f->n_symbols = h.n_symbols;
    memcpy(f->sym, h.sym, 16); memcpy(f->comp, h.comp, 16);
    const char *b = (const char *)base;
    f->contigs.resize(h.n_contigs); f->depths.resize(h.n_contigs); f->exceptions.resize(h.n_exceptions); f->names.resize(h.names_len);
    if (h.n_contigs) { memcpy(f->contigs.data(), b + contigs_at, (size_t)h.n_contigs * sizeof(brx_contig)); memcpy(f->depths.data(), b + depths_at, (size_t)h.n_contigs * sizeof(double)); }
    if (h.n_exceptions) memcpy(f->exceptions.data(), b + exc_at, (size_t)h.n_exceptions * sizeof(brx_exception));
    if (h.names_len) memcpy(f->names.data(), b + names_at, h.names_len);
    f->map_base = base; f->map_len = len;
    f->mapped_packed = (const uint32_t *)(b + at); f->mapped_words = (size_t)h.n_words;
    *out = f;
    return BRX_OK;
}
