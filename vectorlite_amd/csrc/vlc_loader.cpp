// vlc_loader.cpp -- streaming reader of .vlc collection files (the step before
// the hot path).  Reference: load_collection_from_file (src/persistence.rs:149-176) reads the whole
// file into a String, lets serde build CollectionData -- a Vec<Vector> with one heap block per row
// (src/index/flat.rs:59-65) or two HashMaps (src/index/hnsw.rs:272-283) -- and only then validates
// header.version / header.format.
//
// Here the file is mapped, ONE structural pass records where every row's `values` array, `text`
// and `metadata` tokens sit (no number is converted, nothing is copied), the header is validated
// with the reference's error texts, and the numbers are then converted by a pool of host threads
// straight into [chunk, dim] f64 staging blocks that go to the device through the same bulk ingest
// as every other add (k_ingest).  A Vec<Vector> is never materialised; text / metadata stay in the
// file and are handed to the host wrapper as byte ranges.
//
// Number conversion is std::from_chars (correctly rounded, like Rust's str::parse::<f64>).
// serde_json 1.0.145 without its `float_roundtrip` feature (Cargo.toml:27) documents a best-effort
// conversion that can differ in the last bit for long decimal inputs; that difference is unpinned here
// (DESIGN.md section 8).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <charconv>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vectorlite_amd.h"
#include "flat_index.hpp"
#include "hnsw_index.hpp"
#include "vlc_loader.hpp"

namespace vl {
namespace {

struct ParseError {
    std::string msg;
};

struct Range {
    uint64_t off = 0, len = 0;
};

struct RowDesc {
    uint64_t id = 0;
    Range values;  // the `[ ... ]` token
    Range text;    // the string token, quotes included (len 0: absent)
    Range meta;    // any JSON value (len 0: absent = None)
};

class Scanner {
public:
    Scanner(const char* b, const char* e) : base_(b), p_(b), end_(e) {}

    [[noreturn]] void fail(const std::string& what) const
    {
        // serde_json style position: 1-based line, column = bytes since the last newline
        uint64_t line = 1, col = 0;
        for (const char* q = base_; q < p_ && q < end_; ++q) {
            if (*q == '\n') {
                ++line;
                col = 0;
            } else {
                ++col;
            }
        }
        throw ParseError{what + " at line " + std::to_string(line) + " column " + std::to_string(col)};
    }
    void ws()
    {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
    }
    char peek()
    {
        ws();
        if (p_ >= end_) fail("EOF while parsing a value");
        return *p_;
    }
    void expect(char c)
    {
        if (peek() != c) fail(std::string("expected `") + c + "`");
        ++p_;
    }
    bool consume(char c)
    {
        if (peek() == c) {
            ++p_;
            return true;
        }
        return false;
    }
    uint64_t offset() const { return (uint64_t)(p_ - base_); }
    bool at_end()
    {
        ws();
        return p_ >= end_;
    }

    // Skips a string token; returns its range (quotes included).
    Range skip_string()
    {
        if (peek() != '"') fail("expected a string");
        const char* s = p_++;
        for (;;) {
            const char* q = (const char*)memchr(p_, '"', (size_t)(end_ - p_));
            if (!q) {
                p_ = end_;
                fail("EOF while parsing a string");
            }
            // a quote is escaped when preceded by an odd number of backslashes
            const char* b = q;
            while (b > s + 1 && b[-1] == '\\') --b;
            p_ = q + 1;
            if (((q - b) & 1) == 0) break;
        }
        return Range{(uint64_t)(s - base_), (uint64_t)(p_ - s)};
    }
    // Decodes a string token (escapes, \uXXXX with surrogate pairs) into UTF-8.
    std::string string()
    {
        const Range r = skip_string();
        return decode(base_ + r.off, r.len, *this);
    }
    static std::string decode(const char* tok, uint64_t len, const Scanner& ctx)
    {
        std::string out;
        out.reserve(len);
        const char* q = tok + 1;
        const char* e = tok + len - 1;
        auto hex4 = [&](const char* h) -> unsigned {
            unsigned v = 0;
            for (int i = 0; i < 4; ++i) {
                const char c = h[i];
                v <<= 4;
                if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
                else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
                else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
                else ctx.fail("invalid escape");
            }
            return v;
        };
        auto put = [&](unsigned cp) {
            if (cp < 0x80) out.push_back((char)cp);
            else if (cp < 0x800) {
                out.push_back((char)(0xC0 | (cp >> 6)));
                out.push_back((char)(0x80 | (cp & 0x3F)));
            } else if (cp < 0x10000) {
                out.push_back((char)(0xE0 | (cp >> 12)));
                out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                out.push_back((char)(0x80 | (cp & 0x3F)));
            } else {
                out.push_back((char)(0xF0 | (cp >> 18)));
                out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
                out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                out.push_back((char)(0x80 | (cp & 0x3F)));
            }
        };
        while (q < e) {
            const unsigned char c = (unsigned char)*q++;
            if (c < 0x20) ctx.fail("control character (\\u0000-\\u001F) found while parsing a string");
            if (c != '\\') {
                out.push_back((char)c);
                continue;
            }
            if (q >= e) ctx.fail("invalid escape");
            const char esc = *q++;
            switch (esc) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                if (e - q < 4) ctx.fail("invalid escape");
                unsigned cp = hex4(q);
                q += 4;
                if (cp >= 0xD800 && cp < 0xDC00) {
                    if (e - q < 6 || q[0] != '\\' || q[1] != 'u') ctx.fail("unexpected end of hex escape");
                    const unsigned lo = hex4(q + 2);
                    if (lo < 0xDC00 || lo > 0xDFFF) ctx.fail("lone leading surrogate in hex escape");
                    q += 6;
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                } else if (cp >= 0xDC00 && cp < 0xE000) {
                    ctx.fail("lone trailing surrogate in hex escape");
                }
                put(cp);
            } break;
            default: ctx.fail("invalid escape");
            }
        }
        return out;
    }

    // Skips any JSON value; returns its range.  Containers are walked without recursion per element
    // kind (depth bounded like serde_json's recursion limit of 128).
    Range skip_value(int depth = 0)
    {
        if (depth > 128) fail("recursion limit exceeded");
        const char c = peek();
        const char* s = p_;
        if (c == '"') {
            skip_string();
        } else if (c == '{') {
            ++p_;
            if (!consume('}')) {
                for (;;) {
                    skip_string();
                    expect(':');
                    skip_value(depth + 1);
                    if (consume(',')) continue;
                    expect('}');
                    break;
                }
            }
        } else if (c == '[') {
            ++p_;
            if (!consume(']')) {
                for (;;) {
                    skip_value(depth + 1);
                    if (consume(',')) continue;
                    expect(']');
                    break;
                }
            }
        } else if (c == 't') {
            literal("true");
        } else if (c == 'f') {
            literal("false");
        } else if (c == 'n') {
            literal("null");
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            skip_number();
        } else {
            fail("expected value");
        }
        return Range{(uint64_t)(s - base_), (uint64_t)(p_ - s)};
    }
    // Fast skip of an array of numbers (the `values` token): structure is validated later, when the
    // numbers are converted; here only the closing bracket is found.
    Range skip_number_array()
    {
        if (peek() != '[') fail("invalid type: expected a sequence");
        const char* s = p_;
        const char* q = (const char*)memchr(p_, ']', (size_t)(end_ - p_));
        if (!q) {
            p_ = end_;
            fail("EOF while parsing a list");
        }
        p_ = q + 1;
        return Range{(uint64_t)(s - base_), (uint64_t)(p_ - s)};
    }
    void skip_number()
    {
        const char* s = p_;
        if (p_ < end_ && *p_ == '-') ++p_;
        if (p_ >= end_ || *p_ < '0' || *p_ > '9') fail("invalid number");
        if (*p_ == '0') {
            ++p_;
        } else {
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        }
        if (p_ < end_ && *p_ == '.') {
            ++p_;
            if (p_ >= end_ || *p_ < '0' || *p_ > '9') fail("invalid number");
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        }
        if (p_ < end_ && (*p_ == 'e' || *p_ == 'E')) {
            ++p_;
            if (p_ < end_ && (*p_ == '+' || *p_ == '-')) ++p_;
            if (p_ >= end_ || *p_ < '0' || *p_ > '9') fail("invalid number");
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        }
        (void)s;
    }
    uint64_t u64(const char* what)
    {
        const char c = peek();
        if (c < '0' || c > '9') fail(std::string("invalid type: expected ") + what);
        uint64_t v = 0;
        const char* s = p_;
        auto r = std::from_chars(p_, end_, v);
        if (r.ec != std::errc()) fail("number out of range");
        p_ = r.ptr;
        if (p_ < end_ && (*p_ == '.' || *p_ == 'e' || *p_ == 'E')) fail(std::string("invalid type: floating point, expected ") + what);
        if (r.ptr - s > 1 && *s == '0') fail("invalid number");
        return v;
    }

private:
    void literal(const char* lit)
    {
        const size_t n = strlen(lit);
        if ((size_t)(end_ - p_) < n || memcmp(p_, lit, n) != 0) fail("expected ident");
        p_ += n;
    }
    const char* base_;
    const char* p_;
    const char* end_;
};

// Converts the numbers of one `[ ... ]` token into out[0..dim); returns how many it found
// (more than dim are counted, not stored).  JSON number grammar is enforced.
int64_t convert_values(const char* tok, uint64_t len, uint64_t dim, double* out, std::string* err)
{
    const char* p = tok + 1;
    const char* e = tok + len - 1;  // the closing bracket
    auto ws = [&]() {
        while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    };
    int64_t n = 0;
    ws();
    if (p == e) return 0;
    for (;;) {
        ws();
        const char* s = p;
        const char* d = p;
        if (d < e && *d == '-') ++d;
        if (d >= e || *d < '0' || *d > '9') {
            *err = "invalid type: expected f64 in `values`";
            return -1;
        }
        if (*d == '0' && d + 1 < e && d[1] >= '0' && d[1] <= '9') {
            *err = "invalid number in `values`";
            return -1;
        }
        double v = 0.0;
        auto r = std::from_chars(s, e, v);
        if (r.ec == std::errc::result_out_of_range) {
            // from_chars reports overflow AND underflow this way.  serde_json rejects only the former
            // ("number out of range"); 1e-400 is 0.0 and 1e-310 a subnormal there.  strtod tells them apart
            // (correctly rounded in glibc): +-HUGE_VAL means overflow, anything else is the value.
            const std::string tok(s, r.ptr);
            const double t = std::strtod(tok.c_str(), nullptr);
            if (t == HUGE_VAL || t == -HUGE_VAL) {
                *err = "number out of range in `values`";
                return -1;
            }
            v = t;
            r.ec = std::errc();
        }
        if (r.ec != std::errc() || r.ptr == s || r.ptr[-1] == '.' || (r.ptr < e && *r.ptr == '.')) {
            *err = "invalid number in `values`";
            return -1;
        }
        p = r.ptr;
        if ((uint64_t)n < dim) out[n] = v;
        ++n;
        ws();
        if (p == e) return n;
        if (*p != ',') {
            *err = "expected `,` or `]` in `values`";
            return -1;
        }
        ++p;
    }
}

int metric_from_name(const std::string& s)
{
    if (s == "Cosine") return COSINE;
    if (s == "Euclidean") return EUCLIDEAN;
    if (s == "Manhattan") return MANHATTAN;
    if (s == "DotProduct") return DOT;
    return -1;
}

}  // namespace

struct VlcDoc {
    int fd = -1;
    const char* map = nullptr;
    size_t size = 0;
    // header / metadata (src/persistence.rs:88-107)
    std::string version, format, name, index_type_name;
    uint64_t vector_count = 0, dimension = 0;
    // index payload
    int index_type = -1;  // 0 Flat, 1 HNSW
    int metric = -1;      // HNSW only
    uint64_t dim = 0;
    std::vector<RowDesc> rows;
    ~VlcDoc()
    {
        if (map && size) munmap(const_cast<char*>(map), size);
        if (fd >= 0) close(fd);
    }
};

namespace {

// {"id": u64, "values": [...], "text": "...", "metadata": ...}  (struct Vector, src/lib.rs:164-174)
RowDesc scan_flat_row(Scanner& s)
{
    RowDesc r;
    bool has_id = false, has_values = false, has_text = false;
    s.expect('{');
    if (!s.consume('}')) {
        for (;;) {
            const std::string key = s.string();
            s.expect(':');
            if (key == "id") {
                r.id = s.u64("u64");
                has_id = true;
            } else if (key == "values") {
                r.values = s.skip_number_array();
                has_values = true;
            } else if (key == "text") {
                if (s.peek() != '"') s.fail("invalid type: expected a string");
                r.text = s.skip_string();
                has_text = true;
            } else if (key == "metadata") {
                r.meta = s.skip_value();
            } else {
                s.skip_value();  // serde ignores unknown fields
            }
            if (s.consume(',')) continue;
            s.expect('}');
            break;
        }
    }
    if (!has_id) s.fail("missing field `id`");
    if (!has_values) s.fail("missing field `values`");
    if (!has_text) s.fail("missing field `text`");
    return r;
}

void scan_flat(Scanner& s, VlcDoc* d)
{
    bool has_dim = false, has_data = false;
    s.expect('{');
    if (!s.consume('}')) {
        for (;;) {
            const std::string key = s.string();
            s.expect(':');
            if (key == "dim") {
                d->dim = s.u64("usize");
                has_dim = true;
            } else if (key == "data") {
                s.expect('[');
                if (!s.consume(']')) {
                    for (;;) {
                        d->rows.push_back(scan_flat_row(s));
                        if (s.consume(',')) continue;
                        s.expect(']');
                        break;
                    }
                }
                has_data = true;
            } else {
                s.skip_value();
            }
            if (s.consume(',')) continue;
            s.expect('}');
            break;
        }
    }
    if (!has_dim) s.fail("missing field `dim`");
    if (!has_data) s.fail("missing field `data`");
}

// struct Temp { dim, metric, metadata: {id: {text, metadata}}, vector_values: {id: [f64]} }
// (src/index/hnsw.rs:277-283); id_to_index / index_to_id are ignored there and here.
void scan_hnsw(Scanner& s, VlcDoc* d, const char* base)
{
    bool has_dim = false, has_metric = false, has_md = false, has_vv = false;
    struct Side {
        uint64_t id;
        Range text, meta;
    };
    std::vector<Side> side;
    s.expect('{');
    if (!s.consume('}')) {
        for (;;) {
            const std::string key = s.string();
            s.expect(':');
            if (key == "dim") {
                d->dim = s.u64("usize");
                has_dim = true;
            } else if (key == "metric") {
                const std::string m = s.string();
                d->metric = metric_from_name(m);
                if (d->metric < 0) s.fail("unknown variant `" + m + "`, expected one of `Cosine`, `Euclidean`, `Manhattan`, `DotProduct`");
                has_metric = true;
            } else if (key == "vector_values" || key == "metadata") {
                const bool vv = key == "vector_values";
                s.expect('{');
                if (!s.consume('}')) {
                    for (;;) {
                        const std::string idk = s.string();  // map keys are the ids as strings
                        uint64_t id = 0;
                        auto r = std::from_chars(idk.data(), idk.data() + idk.size(), id);
                        if (idk.empty() || r.ec != std::errc() || r.ptr != idk.data() + idk.size()) s.fail("invalid type: expected u64 map key");
                        s.expect(':');
                        if (vv) {
                            RowDesc row;
                            row.id = id;
                            row.values = s.skip_number_array();
                            d->rows.push_back(row);
                        } else {
                            Side sd;
                            sd.id = id;
                            bool has_text = false;
                            s.expect('{');  // struct VectorMetadata { text, metadata }
                            if (!s.consume('}')) {
                                for (;;) {
                                    const std::string k2 = s.string();
                                    s.expect(':');
                                    if (k2 == "text") {
                                        if (s.peek() != '"') s.fail("invalid type: expected a string");
                                        sd.text = s.skip_string();
                                        has_text = true;
                                    } else if (k2 == "metadata") {
                                        sd.meta = s.skip_value();
                                    } else {
                                        s.skip_value();
                                    }
                                    if (s.consume(',')) continue;
                                    s.expect('}');
                                    break;
                                }
                            }
                            if (!has_text) s.fail("missing field `text`");
                            side.push_back(sd);
                        }
                        if (s.consume(',')) continue;
                        s.expect('}');
                        break;
                    }
                }
                (vv ? has_vv : has_md) = true;
            } else {
                s.skip_value();
            }
            if (s.consume(',')) continue;
            s.expect('}');
            break;
        }
    }
    (void)base;
    if (!has_dim) s.fail("missing field `dim`");
    if (!has_metric) s.fail("missing field `metric`");
    if (!has_md) s.fail("missing field `metadata`");
    if (!has_vv) s.fail("missing field `vector_values`");
    if (!side.empty()) {  // attach text / metadata to the rows by id
        std::sort(side.begin(), side.end(), [](const Side& a, const Side& b) { return a.id < b.id; });
        for (RowDesc& r : d->rows) {
            auto it = std::lower_bound(side.begin(), side.end(), r.id, [](const Side& a, uint64_t id) { return a.id < id; });
            if (it != side.end() && it->id == r.id) {
                r.text = it->text;
                r.meta = it->meta;
            }
        }
    }
}

void scan_document(VlcDoc* d)
{
    Scanner s(d->map, d->map + d->size);
    bool has_header = false, has_meta = false, has_index = false;
    s.expect('{');
    if (!s.consume('}')) {
        for (;;) {
            const std::string key = s.string();
            s.expect(':');
            if (key == "header") {  // struct FileHeader (src/persistence.rs:88-93)
                bool v = false, f = false, c = false;
                s.expect('{');
                if (!s.consume('}')) {
                    for (;;) {
                        const std::string k = s.string();
                        s.expect(':');
                        if (k == "version") {
                            d->version = s.string();
                            v = true;
                        } else if (k == "format") {
                            d->format = s.string();
                            f = true;
                        } else if (k == "created_at") {
                            s.string();
                            c = true;
                        } else {
                            s.skip_value();
                        }
                        if (s.consume(',')) continue;
                        s.expect('}');
                        break;
                    }
                }
                if (!v) s.fail("missing field `version`");
                if (!f) s.fail("missing field `format`");
                if (!c) s.fail("missing field `created_at`");
                has_header = true;
            } else if (key == "metadata") {  // struct CollectionMetadata (src/persistence.rs:96-103)
                bool n = false, c = false, vc = false, dm = false, it = false;
                s.expect('{');
                if (!s.consume('}')) {
                    for (;;) {
                        const std::string k = s.string();
                        s.expect(':');
                        if (k == "name") {
                            d->name = s.string();
                            n = true;
                        } else if (k == "created_at") {
                            s.string();
                            c = true;
                        } else if (k == "vector_count") {
                            d->vector_count = s.u64("usize");
                            vc = true;
                        } else if (k == "dimension") {
                            d->dimension = s.u64("usize");
                            dm = true;
                        } else if (k == "index_type") {
                            d->index_type_name = s.string();
                            it = true;
                        } else {
                            s.skip_value();
                        }
                        if (s.consume(',')) continue;
                        s.expect('}');
                        break;
                    }
                }
                if (!n) s.fail("missing field `name`");
                if (!c) s.fail("missing field `created_at`");
                if (!vc) s.fail("missing field `vector_count`");
                if (!dm) s.fail("missing field `dimension`");
                if (!it) s.fail("missing field `index_type`");
                has_meta = true;
            } else if (key == "index") {  // externally tagged enum VectorIndexWrapper (src/lib.rs:270-276)
                s.expect('{');
                const std::string variant = s.string();
                s.expect(':');
                if (variant == "Flat") {
                    d->index_type = 0;
                    scan_flat(s, d);
                } else if (variant == "HNSW") {
                    d->index_type = 1;
                    scan_hnsw(s, d, d->map);
                } else {
                    s.fail("unknown variant `" + variant + "`, expected `Flat` or `HNSW`");
                }
                s.expect('}');
                has_index = true;
            } else {
                s.skip_value();
            }
            if (s.consume(',')) continue;
            s.expect('}');
            break;
        }
    }
    if (!has_header) s.fail("missing field `header`");
    if (!has_meta) s.fail("missing field `metadata`");
    if (!has_index) s.fail("missing field `index`");
    if (!s.at_end()) s.fail("trailing characters");
}

unsigned worker_count(uint64_t rows)
{
    unsigned n = std::thread::hardware_concurrency();
    if (const char* e = getenv("VL_VLC_THREADS"))
        if (*e) n = (unsigned)atoi(e);
    if (n < 1) n = 1;
    if (n > 32) n = 32;
    if (rows < 256) n = 1;
    return n;
}

}  // namespace

int vlc_open(const char* path, VlcDoc** out)
{
    if (!path || !out) return ERR_INVALID_ARG;
    *out = nullptr;
    std::unique_ptr<VlcDoc> d(new VlcDoc());
    d->fd = open(path, O_RDONLY | O_CLOEXEC);
    if (d->fd < 0) {
        if (errno == ENOENT) {
            set_last_error(std::string("File not found: ") + path);  // PersistenceError::FileNotFound
            return VL_ERR_FILE_NOT_FOUND;
        }
        set_last_error(std::string("IO error: ") + strerror(errno));
        return VL_ERR_IO;
    }
    struct stat st;
    if (fstat(d->fd, &st) != 0) {
        set_last_error(std::string("IO error: ") + strerror(errno));
        return VL_ERR_IO;
    }
    d->size = (size_t)st.st_size;
    if (d->size) {
        void* m = mmap(nullptr, d->size, PROT_READ, MAP_PRIVATE, d->fd, 0);
        if (m == MAP_FAILED) {
            d->size = 0;
            set_last_error(std::string("IO error: ") + strerror(errno));
            return VL_ERR_IO;
        }
        d->map = (const char*)m;
        (void)madvise(m, d->size, MADV_SEQUENTIAL);
    }
    try {
        scan_document(d.get());
    } catch (const ParseError& e) {
        set_last_error("Serialization error: " + e.msg);  // PersistenceError::Serialization
        return VL_ERR_SERIALIZATION;
    }
    // the reference validates after serde has accepted the document (src/persistence.rs:160-173)
    if (d->version != "1.0.0") {
        set_last_error("Version mismatch: expected 1.0.0, got " + d->version);
        return VL_ERR_VERSION_MISMATCH;
    }
    if (d->format != "vectorlite-collection") {
        set_last_error("Invalid file format: Expected format 'vectorlite-collection', got '" + d->format + "'");
        return VL_ERR_INVALID_FORMAT;
    }
    if (d->index_type == 1 && d->dim == 0) {  // src/index/hnsw.rs:288-290 (a serde custom error)
        set_last_error("Serialization error: Invalid dimension: cannot be 0");
        return VL_ERR_SERIALIZATION;
    }
    *out = d.release();
    return OK;
}

void vlc_close(VlcDoc* d) { delete d; }

const char* vlc_name(const VlcDoc* d) { return d->name.c_str(); }

void vlc_info(const VlcDoc* d, int* index_type, int* metric, uint64_t* dim, uint64_t* rows, uint64_t* vector_count,
              uint64_t* dimension)
{
    if (index_type) *index_type = d->index_type;
    if (metric) *metric = d->metric;
    if (dim) *dim = d->dim;
    if (rows) *rows = d->rows.size();
    if (vector_count) *vector_count = d->vector_count;
    if (dimension) *dimension = d->dimension;
}

int vlc_side_table(const VlcDoc* d, uint64_t* ids, uint64_t* text_off, uint64_t* text_len, uint64_t* meta_off,
                   uint64_t* meta_len)
{
    for (size_t i = 0; i < d->rows.size(); ++i) {
        const RowDesc& r = d->rows[i];
        if (ids) ids[i] = r.id;
        if (text_off) text_off[i] = r.text.off;
        if (text_len) text_len[i] = r.text.len;
        if (meta_off) meta_off[i] = r.meta.off;
        if (meta_len) meta_len[i] = r.meta.len;
    }
    return OK;
}

// rows [first, first + n) -> out[n, dim]; host threads split the range.
int vlc_read_values(const VlcDoc* d, uint64_t first, uint64_t n, double* out)
{
    if (first > d->rows.size() || n > d->rows.size() - first) return ERR_INVALID_ARG;
    if (n == 0) return OK;
    if (!out && d->dim) return ERR_INVALID_ARG;
    const uint64_t dim = d->dim;
    const unsigned nt = worker_count(n);
    std::atomic<uint64_t> next{0};
    std::atomic<bool> failed{false};
    std::string err;
    std::mutex err_mu;
    auto work = [&]() {
        const uint64_t step = 64;
        for (;;) {
            const uint64_t b = next.fetch_add(step);
            if (b >= n || failed.load(std::memory_order_relaxed)) return;
            const uint64_t e = std::min(n, b + step);
            for (uint64_t i = b; i < e; ++i) {
                const RowDesc& r = d->rows[first + i];
                std::string msg;
                const int64_t got = convert_values(d->map + r.values.off, r.values.len, dim, out + i * dim, &msg);
                if (got < 0 || (uint64_t)got != dim) {
                    if (got >= 0) {
                        // a Flat payload with a ragged row deserialises in the reference and panics at the first
                        // search (assert_eq! in calculate, src/lib.rs:382); the slab needs uniform rows, so it is
                        // refused here.  HNSW: src/index/hnsw.rs:325-330.
                        msg = "Vector dimension mismatch: expected " + std::to_string(dim) + ", got " + std::to_string(got);
                    }
                    std::lock_guard<std::mutex> g(err_mu);
                    if (!failed.exchange(true)) err = msg + " (row " + std::to_string(first + i) + ", id " + std::to_string(r.id) + ")";
                    return;
                }
            }
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (failed.load()) {
        set_last_error("Serialization error: " + err);
        return VL_ERR_SERIALIZATION;
    }
    return OK;
}

// Builds the GPU index: staging blocks of at most ~256 MB are converted on the host threads and
// ingested on the device while the next block is being converted.
int vlc_build_index(const VlcDoc* d, int device, GpuFlatIndex** out_flat, HnswIndex** out_hnsw)
{
    *out_flat = nullptr;
    *out_hnsw = nullptr;
    const uint64_t n = d->rows.size(), dim = d->dim;
    std::unique_ptr<GpuFlatIndex> flat;
    std::unique_ptr<HnswIndex> hnsw;
    if (d->index_type == 0) {
        GpuFlatIndex* f = nullptr;
        int rc = GpuFlatIndex::create(dim, device, &f);
        if (rc != OK) return rc;
        flat.reset(f);
        if (n) {
            rc = flat->reserve(n);
            if (rc != OK) return rc;
        }
    } else {
        HnswIndex* h = nullptr;
        int rc = HnswIndex::create(dim, d->metric, HnswParams(), device, &h);
        if (rc != OK) return rc;
        hnsw.reset(h);
    }
    if (n) {
        const uint64_t row_bytes = std::max<uint64_t>(dim, 1) * sizeof(double);
        uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / row_bytes);
        chunk = std::min<uint64_t>(chunk, n);
        std::vector<double> stage[2];
        stage[0].resize(chunk * dim);
        if (chunk < n) stage[1].resize(chunk * dim);
        std::vector<uint64_t> ids(chunk);
        int rc = vlc_read_values(d, 0, chunk, stage[0].data());
        if (rc != OK) return rc;
        int cur = 0;
        for (uint64_t first = 0; first < n; first += chunk) {
            const uint64_t m = std::min(chunk, n - first);
            const uint64_t nfirst = first + m, nm = nfirst < n ? std::min(chunk, n - nfirst) : 0;
            int rc_next = OK;
            std::string next_err;
            std::thread prefetch;
            if (nm) {
                prefetch = std::thread([&, nfirst, nm, cur]() {
                    rc_next = vlc_read_values(d, nfirst, nm, stage[cur ^ 1].data());
                    if (rc_next != OK) next_err = last_error();  // thread-local in the worker: carry it over
                });
            }
            for (uint64_t i = 0; i < m; ++i) ids[i] = d->rows[first + i].id;
            // Flat: FlatIndex{dim, data} is filled with no validation (duplicate ids are kept,
            // src/index/flat.rs:59); HNSW re-inserts every vector (src/index/hnsw.rs:320-347).
            rc = flat ? flat->add_bulk(ids.data(), stage[cur].data(), m, /*validate=*/false, /*values_on_device=*/false)
                      : hnsw->add_bulk(ids.data(), stage[cur].data(), m, /*values_on_device=*/false);
            if (prefetch.joinable()) prefetch.join();
            if (rc != OK) return rc;
            if (rc_next != OK) {
                set_last_error(next_err);
                return rc_next;
            }
            cur ^= 1;
        }
    }
    *out_flat = flat.release();
    *out_hnsw = hnsw.release();
    return OK;
}

}  // namespace vl
