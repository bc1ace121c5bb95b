// mb_seq.cpp -- FASTA ingestion for libmiblast: what lastz does when handed
// `file.fa[multiple][nameparse=darkspace]` (/root/reference/src/cactus/paf/local_alignment.py:60-62):
// every record becomes one contig of a single concatenated sequence, names stop at the first
// blank.  Bases are encoded to one code byte each (0..3 ACGT, 4 N/IUPAC, bit 3 = soft-masked) --
// the alphabet is ACGTNacgtn after cactus_sanitizeFastaHeaders
// (/root/reference/preprocessor/cactus_sanitizeFastaHeaders.c:35-49,148-160).
#include "mb_common.h"

#include <algorithm>
#include <array>
#include <cstring>

namespace mb {

namespace {
std::array<uint8_t, 256> make_code_table() {
    std::array<uint8_t, 256> t{};
    for (int c = 0; c < 256; c++) t[c] = (c >= 'a' && c <= 'z') ? 12 : 4;
    t['A'] = 0; t['C'] = 1; t['G'] = 2; t['T'] = 3;
    t['a'] = 8; t['c'] = 9; t['g'] = 10; t['t'] = 11;
    return t;
}
const std::array<uint8_t, 256> kCode = make_code_table();
inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }
}  // namespace

int SeqSet::contig_of(int64_t pos) const {
    auto it = std::upper_bound(starts.begin(), starts.end(), pos);
    return (int)(it - starts.begin()) - 1;
}

int parse_fasta(const char *buf, size_t len, SeqSet &out) {
    out.names.clear(); out.starts.clear(); out.lens.clear();
    out.codes.clear();
    out.codes.reserve(len + 2);
    out.codes.push_back(kSep);                        // position -1
    const char *p = buf, *end = buf + len;
    bool open = false;
    auto close_record = [&]() {
        if (open && out.lens.back() == 0) {           // empty record: drop it (and the separator added for it)
            out.names.pop_back(); out.starts.pop_back(); out.lens.pop_back();
            if (!out.names.empty()) out.codes.pop_back();
        }
    };
    while (p < end) {
        const char *eol = (const char *)memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        if (*p == '>') {
            close_record();
            const char *n0 = p + 1, *n1 = n0;
            while (n1 < eol && !is_blank(*n1)) n1++;
            if (!out.names.empty()) out.codes.push_back(kSep);
            out.names.emplace_back(n0, n1);
            out.starts.push_back((int64_t)out.codes.size() - 1);
            out.lens.push_back(0);
            open = true;
        } else if (open) {
            int64_t added = 0;
            for (const char *c = p; c < eol; c++) {
                if (is_blank(*c)) continue;
                out.codes.push_back(kCode[(unsigned char)*c]);
                added++;
            }
            out.lens.back() += added;
        }
        p = eol + 1;
    }
    close_record();
    out.total = (int64_t)out.codes.size() - 1;
    out.codes.push_back(kSep);                        // position total
    return 0;
}

}  // namespace mb
