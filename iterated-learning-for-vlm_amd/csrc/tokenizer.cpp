// Host-side byte-level BPE tokenizer of the text tower (no GPU work): replaces the per-step Python BPE the reference
// runs on the training thread inside forward() -- prototype/model/utils/text_utils/simple_tokenizer.py:63-135
// (vocabulary, merge loop, pre-tokenisation regex) and text_encoder/text_transformer.py:155-202 (SOT/EOT framing,
// truncation, pad mask).  Vocabulary: 256 byte symbols, 256 end-of-word byte symbols, 48894 merges, <|mask|>,
// <|startoftext|>, <|endoftext|> (49409 ids).
//
// Scope of the native path: captions made of printable ASCII and ASCII white space without '&'.  For those the
// reference's cleaning steps (ftfy, html.unescape, Unicode-aware regex classes, str.lower) reduce to byte rules that
// are restated here exactly.  Any other caption is flagged in `fallback[i]` and left untouched, and the Python caller
// tokenises it with its own full-Unicode implementation (both are pinned by tests/golden/g9_tokenizer.json).
#include <cstdint>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ilvlm_hip.h"

// error plumbing shared with the HIP translation units (api.hip); this file is plain host C++
void ilvlm_set_error(const char* fmt, ...);
#define ILVLM_FAIL(code, ...)          \
    do {                               \
        ilvlm_set_error(__VA_ARGS__);  \
        return (code);                 \
    } while (0)
#define ILVLM_REQUIRE(cond, ...) \
    do {                         \
        if (!(cond)) ILVLM_FAIL(ILVLM_ERR_ARG, __VA_ARGS__); \
    } while (0)

namespace {

constexpr int N_MERGES = 49152 - 256 - 2;
constexpr int ID_SOT = 49407, ID_EOT = 49408;

struct PairHash {
    size_t operator()(uint64_t k) const { return (size_t)(k * 0x9E3779B97F4A7C15ull >> 17); }
};

struct Tokenizer {
    // internal symbol id -> vocabulary id (the LAST vocabulary entry with that string, as a Python dict built by
    // enumerate() keeps it)
    std::vector<int> out_id;
    std::unordered_map<std::string, int> intern;                       // symbol string -> internal id
    std::unordered_map<uint64_t, std::pair<int, int>, PairHash> merge;  // (a, b) -> (rank, merged internal id)
    int byte_sym[256];      // byte -> internal id of its symbol
    int byte_sym_eow[256];  // byte -> internal id of symbol + "</w>"
    int sot_sym, eot_sym;
    std::unordered_map<std::string, std::vector<int>> cache;           // pre-token -> vocabulary ids
    std::mutex mu;

    int intern_symbol(const std::string& s, int vocab_id) {
        auto it = intern.find(s);
        if (it == intern.end()) {
            int id = (int)out_id.size();
            intern.emplace(s, id);
            out_id.push_back(vocab_id);
            return id;
        }
        out_id[it->second] = vocab_id;   // later duplicate wins
        return it->second;
    }
};

void append_utf8(std::string& s, int cp) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 63))); }
    else { s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 63))); s.push_back((char)(0x80 | (cp & 63))); }
}

// GPT-2 byte -> printable code point table; insertion order (kept bytes first, then the rest) defines ids 0..255
void byte_order(int order[256], int codepoint[256]) {
    bool keep[256] = {false};
    int n = 0;
    auto add_range = [&](int lo, int hi) { for (int b = lo; b <= hi; ++b) { keep[b] = true; order[n++] = b; codepoint[b] = b; } };
    add_range('!', '~');
    add_range(0xA1, 0xAC);
    add_range(0xAE, 0xFF);
    int extra = 0;
    for (int b = 0; b < 256; ++b)
        if (!keep[b]) { order[n++] = b; codepoint[b] = 256 + extra++; }
}

inline bool is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }
inline bool is_letter(unsigned char c) { return c >= 'a' && c <= 'z'; }   // after lower-casing
inline bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }

// length of the pre-token starting at s[i] (0 = white space: skip one byte).  Alternatives in the reference's order.
size_t match_piece(const std::string& s, size_t i) {
    static const char* specials[2] = {"<|startoftext|>", "<|endoftext|>"};
    for (const char* sp : specials) {
        size_t n = std::strlen(sp);
        if (s.compare(i, n, sp) == 0) return n;
    }
    if (s[i] == '\'') {
        static const char* contr[7] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};
        for (const char* c : contr) {
            size_t n = std::strlen(c);
            if (s.compare(i, n, c) == 0) return n;
        }
    }
    unsigned char c = (unsigned char)s[i];
    if (is_letter(c)) {
        size_t j = i + 1;
        while (j < s.size() && is_letter((unsigned char)s[j])) ++j;
        return j - i;
    }
    if (is_digit(c)) return 1;
    if (is_space(c)) return 0;
    size_t j = i + 1;
    while (j < s.size()) {
        unsigned char d = (unsigned char)s[j];
        if (is_space(d) || is_letter(d) || is_digit(d)) break;
        ++j;
    }
    return j - i;
}

// greedy lowest-rank pair merging of one pre-token (internal ids in, vocabulary ids out)
void merge_word(const Tokenizer& t, std::vector<int>& word, std::vector<int>& out) {
    while (word.size() > 1) {
        int best_rank = std::numeric_limits<int>::max(), best_a = -1, best_b = -1, best_m = -1;
        for (size_t i = 0; i + 1 < word.size(); ++i) {
            auto it = t.merge.find(((uint64_t)(uint32_t)word[i] << 32) | (uint32_t)word[i + 1]);
            if (it != t.merge.end() && it->second.first < best_rank) {
                best_rank = it->second.first; best_a = word[i]; best_b = word[i + 1]; best_m = it->second.second;
            }
        }
        if (best_a < 0) break;
        size_t w = 0;
        for (size_t i = 0; i < word.size();) {
            if (i + 1 < word.size() && word[i] == best_a && word[i + 1] == best_b) { word[w++] = best_m; i += 2; }
            else word[w++] = word[i++];
        }
        word.resize(w);
    }
    for (int s : word) out.push_back(t.out_id[s]);
}

}  // namespace

extern "C" int ilvlm_tokenizer_create(const char* merges_text, long nbytes, void** handle) {
    ILVLM_REQUIRE(merges_text && handle && nbytes > 0, "tokenizer_create: null or empty vocabulary text");
    Tokenizer* t = new Tokenizer();
    int order[256], cp[256];
    byte_order(order, cp);
    std::vector<std::string> sym(256);
    for (int i = 0; i < 256; ++i) {
        std::string s;
        append_utf8(s, cp[order[i]]);
        sym[order[i]] = s;
        t->byte_sym[order[i]] = t->intern_symbol(s, i);
    }
    for (int i = 0; i < 256; ++i) t->byte_sym_eow[order[i]] = t->intern_symbol(sym[order[i]] + "</w>", 256 + i);
    // merges: line 0 is the version header; one "left right" pair per line
    const char* p = merges_text;
    const char* end = merges_text + nbytes;
    auto next_line = [&](std::string& line) -> bool {
        if (p >= end) return false;
        const char* q = (const char*)memchr(p, '\n', (size_t)(end - p));
        if (!q) q = end;
        line.assign(p, q);
        p = q + 1;
        return true;
    };
    std::string line;
    next_line(line);
    int rank = 0;
    while (rank < N_MERGES && next_line(line)) {
        size_t sp = line.find(' ');
        if (sp == std::string::npos || sp == 0 || sp + 1 >= line.size()) {
            delete t;
            ILVLM_FAIL(ILVLM_ERR_ARG, "tokenizer_create: malformed merge line %d", rank + 1);
        }
        std::string a = line.substr(0, sp), b = line.substr(sp + 1);
        while (!b.empty() && (b.back() == '\r' || b.back() == ' ')) b.pop_back();
        auto ia = t->intern.find(a), ib = t->intern.find(b);
        if (ia == t->intern.end() || ib == t->intern.end()) {
            delete t;
            ILVLM_FAIL(ILVLM_ERR_ARG, "tokenizer_create: merge %d uses a symbol that no earlier merge produced", rank);
        }
        int m = t->intern_symbol(a + b, 512 + rank);
        uint64_t key = ((uint64_t)(uint32_t)ia->second << 32) | (uint32_t)ib->second;
        t->merge[key] = std::make_pair(rank, m);   // a repeated pair keeps its last rank, as the reference's dict does
        ++rank;
    }
    if (rank != N_MERGES) {
        delete t;
        ILVLM_FAIL(ILVLM_ERR_ARG, "tokenizer_create: %d merges in the vocabulary text, expected %d", rank, N_MERGES);
    }
    t->intern_symbol("<|mask|>", 512 + N_MERGES);
    t->sot_sym = t->intern_symbol("<|startoftext|>", ID_SOT);
    t->eot_sym = t->intern_symbol("<|endoftext|>", ID_EOT);
    *handle = t;
    return ILVLM_OK;
}

extern "C" int ilvlm_tokenizer_destroy(void* handle) {
    delete (Tokenizer*)handle;
    return ILVLM_OK;
}

extern "C" int ilvlm_tokenizer_encode(void* handle, const char* const* texts, int n, int context_length, long long* tokens,
                                      float* pad_mask, int* lengths, unsigned char* fallback) {
    ILVLM_REQUIRE(handle && texts && tokens && pad_mask && lengths && fallback, "tokenizer_encode: null pointer");
    ILVLM_REQUIRE(n >= 0 && context_length >= 2, "tokenizer_encode: bad n=%d context_length=%d", n, context_length);
    Tokenizer& t = *(Tokenizer*)handle;
    std::lock_guard<std::mutex> lock(t.mu);
    const float ninf = -std::numeric_limits<float>::infinity();
    std::vector<int> ids, word;
    std::string clean, key;
    for (int i = 0; i < n; ++i) {
        const char* src = texts[i];
        ILVLM_REQUIRE(src != nullptr, "tokenizer_encode: texts[%d] is null", i);
        long long* trow = tokens + (size_t)i * context_length;
        float* mrow = pad_mask + (size_t)i * context_length;
        bool native = true;
        for (const unsigned char* c = (const unsigned char*)src; *c; ++c)
            if (*c >= 0x7F || *c == '&' || (*c < 0x20 && !is_space(*c))) { native = false; break; }
        fallback[i] = native ? 0 : 1;
        if (!native) { lengths[i] = 0; continue; }
        // strip, collapse white-space runs to one blank, lower-case
        clean.clear();
        bool pending = false;
        for (const unsigned char* c = (const unsigned char*)src; *c; ++c) {
            if (is_space(*c)) { pending = !clean.empty(); continue; }
            if (pending) { clean.push_back(' '); pending = false; }
            clean.push_back((*c >= 'A' && *c <= 'Z') ? (char)(*c + 32) : (char)*c);
        }
        ids.clear();
        ids.push_back(ID_SOT);
        for (size_t pos = 0; pos < clean.size();) {
            size_t len = match_piece(clean, pos);
            if (len == 0) { ++pos; continue; }
            key.assign(clean, pos, len);
            pos += len;
            auto hit = t.cache.find(key);
            if (hit == t.cache.end()) {
                std::vector<int> out;
                if (key == "<|startoftext|>") out.push_back(ID_SOT);
                else if (key == "<|endoftext|>") out.push_back(ID_EOT);
                else {
                    word.clear();
                    for (size_t k = 0; k + 1 < key.size(); ++k) word.push_back(t.byte_sym[(unsigned char)key[k]]);
                    word.push_back(t.byte_sym_eow[(unsigned char)key.back()]);
                    merge_word(t, word, out);
                }
                hit = t.cache.emplace(key, std::move(out)).first;
            }
            ids.insert(ids.end(), hit->second.begin(), hit->second.end());
        }
        ids.push_back(ID_EOT);
        int len = (int)ids.size();
        if (len > context_length) {      // keep [sot] + tok[1 : ctx-1] + [eot]
            ids[context_length - 1] = ids.back();
            len = context_length;
        }
        for (int k = 0; k < context_length; ++k) {
            trow[k] = k < len ? ids[k] : 0;
            mrow[k] = k < len ? 0.f : ninf;
        }
        lengths[i] = len;
    }
    return ILVLM_OK;
}
