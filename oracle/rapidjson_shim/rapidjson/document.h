// TEST INFRASTRUCTURE ONLY (oracle/): a from-scratch, minimal stand-in for the subset of the
// rapidjson API that the reference engine uses (reference: extern/rapidjson is an EMPTY un-vendored
// submodule, .gitmodules:1-3, pinned commit unknown).  It exists solely so that the UNMODIFIED
// reference sources under /root/reference/src compile into oracle/_ref/ (see oracle/Makefile).
// Nothing in the product path (cityflow_amd/) includes this header.
//
// Numeric note: decimal -> double conversion follows real rapidjson's DEFAULT number reader (not a
// correctly rounded strtod: it can be an ulp or two off on 16/17-digit literals) as restated in number_reader.h from the
// library's published algorithm; the product's host parser has its own restatement (csrc/host/json_number.h) and
// oracle/probe_json_number.cpp compares the two.  What stays unpinned: the library itself is not here to run.
#ifndef ORACLE_RAPIDJSON_SHIM_DOCUMENT_H
#define ORACLE_RAPIDJSON_SHIM_DOCUMENT_H

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <climits>
#include <string>
#include <vector>
#include <utility>

#include "number_reader.h"

namespace rapidjson {

typedef unsigned SizeType;

enum Type { kNullType = 0, kFalseType, kTrueType, kObjectType, kArrayType, kStringType, kNumberType };

enum ParseErrorCode { kParseErrorNone = 0, kParseErrorSyntax = 1 };

class CrtAllocator {};
template <typename Base = CrtAllocator> class MemoryPoolAllocator {};

struct StringRefType {
    const char *s;
    explicit StringRefType(const char *str) : s(str) {}
};
inline StringRefType StringRef(const char *str) { return StringRefType(str); }

class Value;
struct Member;

class Value {
public:
    typedef MemoryPoolAllocator<> AllocatorType;
    typedef const Member *ConstMemberIterator;
    typedef Member *MemberIterator;

    // ---- construction (move-only, like rapidjson) ----
    Value() : kind_(kNullType), isInt_(false), num_(0), int_(0) {}
    explicit Value(Type t) : kind_(t), isInt_(false), num_(0), int_(0) {}
    Value(const std::string &s, AllocatorType &) : kind_(kStringType), isInt_(false), num_(0), int_(0), str_(s) {}
    Value(const char *s, AllocatorType &) : kind_(kStringType), isInt_(false), num_(0), int_(0), str_(s) {}
    explicit Value(bool b) : kind_(b ? kTrueType : kFalseType), isInt_(false), num_(0), int_(0) {}
    explicit Value(int v) : kind_(kNumberType), isInt_(true), num_((double) v), int_(v) {}
    explicit Value(unsigned v) : kind_(kNumberType), isInt_(true), num_((double) v), int_(v) {}
    explicit Value(short v) : kind_(kNumberType), isInt_(true), num_((double) v), int_(v) {}
    explicit Value(int64_t v) : kind_(kNumberType), isInt_(true), num_((double) v), int_(v) {}
    explicit Value(double v) : kind_(kNumberType), isInt_(false), num_(v), int_(0) {}
    Value(const Value &) = delete;
    Value &operator=(const Value &) = delete;
    Value(Value &&o) noexcept { moveFrom(o); }
    Value &operator=(Value &&o) noexcept {
        if (this != &o) moveFrom(o);
        return *this;
    }
    ~Value();

    Value &Move() { return *this; }

    // ---- type queries ----
    bool IsNull() const { return kind_ == kNullType; }
    bool IsBool() const { return kind_ == kTrueType || kind_ == kFalseType; }
    bool IsObject() const { return kind_ == kObjectType; }
    bool IsArray() const { return kind_ == kArrayType; }
    bool IsString() const { return kind_ == kStringType; }
    bool IsNumber() const { return kind_ == kNumberType; }
    bool IsInt() const { return kind_ == kNumberType && isInt_ && int_ >= INT_MIN && int_ <= INT_MAX; }
    bool IsUint() const { return kind_ == kNumberType && isInt_ && int_ >= 0 && int_ <= (int64_t) UINT_MAX; }
    bool IsDouble() const { return kind_ == kNumberType && !isInt_; }

    template <typename T> bool Is() const;
    template <typename T> T Get() const;

    bool GetBool() const { return kind_ == kTrueType; }
    int GetInt() const { return (int) int_; }
    unsigned GetUint() const { return (unsigned) int_; }
    double GetDouble() const { return isInt_ ? (double) int_ : num_; }
    const char *GetString() const { return str_.c_str(); }

    // ---- arrays ----
    SizeType Size() const { return (SizeType) arr_.size(); }
    bool Empty() const { return arr_.empty(); }
    Value &operator[](SizeType i) { return arr_[i]; }
    const Value &operator[](SizeType i) const { return arr_[i]; }

    struct ArrayRange {
        Value *b, *e;
        Value *begin() const { return b; }
        Value *end() const { return e; }
    };
    struct ConstArrayRange {
        const Value *b, *e;
        const Value *begin() const { return b; }
        const Value *end() const { return e; }
    };
    ArrayRange GetArray() { return ArrayRange{arr_.data(), arr_.data() + arr_.size()}; }
    ConstArrayRange GetArray() const { return ConstArrayRange{arr_.data(), arr_.data() + arr_.size()}; }

    Value &PushBack(Value &v, AllocatorType &) {
        arr_.emplace_back(std::move(v));
        return *this;
    }
    Value &PushBack(Value &&v, AllocatorType &) {
        arr_.emplace_back(std::move(v));
        return *this;
    }
    Value &PushBack(double v, AllocatorType &) {
        arr_.emplace_back(Value(v));
        return *this;
    }
    Value &PushBack(int v, AllocatorType &) {
        arr_.emplace_back(Value(v));
        return *this;
    }

    // ---- objects ----
    ConstMemberIterator FindMember(const char *name) const;
    ConstMemberIterator MemberBegin() const;
    ConstMemberIterator MemberEnd() const;
    bool HasMember(const char *name) const { return FindMember(name) != MemberEnd(); }

    Value &SetObject() {
        clear();
        kind_ = kObjectType;
        return *this;
    }
    Value &SetArray() {
        clear();
        kind_ = kArrayType;
        return *this;
    }
    Value &SetString(StringRefType s) {
        clear();
        kind_ = kStringType;
        str_ = s.s;
        return *this;
    }
    Value &SetString(const char *s, AllocatorType &) {
        clear();
        kind_ = kStringType;
        str_ = s;
        return *this;
    }
    // string-literal form used as SetString("null")
    template <size_t N> Value &SetString(const char (&s)[N]) {
        clear();
        kind_ = kStringType;
        str_ = s;
        return *this;
    }

    // AddMember: name may be a string literal or a Value; value may be a Value (moved from) or a scalar
    template <size_t N> Value &AddMember(const char (&name)[N], Value &v, AllocatorType &a) {
        Value n(name, a);
        return addMemberImpl(n, v);
    }
    template <size_t N> Value &AddMember(const char (&name)[N], Value &&v, AllocatorType &a) {
        Value n(name, a);
        return addMemberImpl(n, v);
    }
    template <size_t N, typename T> Value &AddMember(const char (&name)[N], T v, AllocatorType &a) {
        Value n(name, a);
        Value val(v);
        return addMemberImpl(n, val);
    }
    Value &AddMember(Value &name, Value &v, AllocatorType &) { return addMemberImpl(name, v); }
    Value &AddMember(Value &name, Value &&v, AllocatorType &) { return addMemberImpl(name, v); }
    Value &AddMember(Value &&name, Value &v, AllocatorType &) { return addMemberImpl(name, v); }
    Value &AddMember(Value &&name, Value &&v, AllocatorType &) { return addMemberImpl(name, v); }

    // ---- serialisation helper (used by Writer) ----
    void writeTo(std::string &out) const;

protected:
    friend class Document;
    friend struct Parser;
    void clear();
    void moveFrom(Value &o);
    Value &addMemberImpl(Value &name, Value &v);

    Type kind_;
    bool isInt_;
    double num_;
    int64_t int_;
    std::string str_;
    std::vector<Value> arr_;
    std::vector<Member> *obj_ = nullptr;  // pointer because Member is incomplete here
    mutable size_t findHint_ = 0;         // where FindMember looks first
};

struct Member {
    Value name;
    Value value;
    Member() {}
    Member(Member &&o) noexcept : name(std::move(o.name)), value(std::move(o.value)) {}
    Member &operator=(Member &&o) noexcept {
        name = std::move(o.name);
        value = std::move(o.value);
        return *this;
    }
};

inline Value::~Value() { delete obj_; }

inline void Value::clear() {
    str_.clear();
    arr_.clear();
    delete obj_;
    obj_ = nullptr;
    isInt_ = false;
    num_ = 0;
    int_ = 0;
    kind_ = kNullType;
}

inline void Value::moveFrom(Value &o) {
    delete obj_;
    kind_ = o.kind_;
    isInt_ = o.isInt_;
    num_ = o.num_;
    int_ = o.int_;
    str_ = std::move(o.str_);
    arr_ = std::move(o.arr_);
    obj_ = o.obj_;
    o.obj_ = nullptr;
    o.kind_ = kNullType;
}

inline Value &Value::addMemberImpl(Value &name, Value &v) {
    if (!obj_) obj_ = new std::vector<Member>();
    obj_->emplace_back();
    obj_->back().name = std::move(name);
    obj_->back().value = std::move(v);
    return *this;
}

inline Value::ConstMemberIterator Value::MemberBegin() const {
    static const std::vector<Member> empty;
    const std::vector<Member> &m = obj_ ? *obj_ : empty;
    return m.data();
}
inline Value::ConstMemberIterator Value::MemberEnd() const {
    static const std::vector<Member> empty;
    const std::vector<Member> &m = obj_ ? *obj_ : empty;
    return m.data() + m.size();
}
// (same result as the library's linear search — member names are unique in the files this reads — but the search starts
// behind the member found last: Archive's loader asks for an object's members in file order, and a 100x100 grid's archive has
// 481 k drivables in one object)
inline Value::ConstMemberIterator Value::FindMember(const char *name) const {
    ConstMemberIterator b = MemberBegin(), e = MemberEnd();
    const size_t n = (size_t) (e - b);
    size_t at = findHint_ < n ? findHint_ : 0;
    for (size_t k = 0; k < n; ++k) {
        if ((b + at)->name.str_ == name) {
            findHint_ = at + 1;
            return b + at;
        }
        if (++at == n) at = 0;
    }
    return e;
}

template <> inline bool Value::Is<bool>() const { return IsBool(); }
template <> inline bool Value::Is<int>() const { return IsInt(); }
template <> inline bool Value::Is<unsigned>() const { return IsUint(); }
template <> inline bool Value::Is<double>() const { return IsNumber(); }
template <> inline bool Value::Is<const char *>() const { return IsString(); }
template <> inline bool Value::Get<bool>() const { return GetBool(); }
template <> inline int Value::Get<int>() const { return GetInt(); }
template <> inline unsigned Value::Get<unsigned>() const { return GetUint(); }
template <> inline double Value::Get<double>() const { return GetDouble(); }
template <> inline const char *Value::Get<const char *>() const { return GetString(); }

inline void Value::writeTo(std::string &out) const {
    char buf[64];
    switch (kind_) {
        case kNullType: out += "null"; break;
        case kFalseType: out += "false"; break;
        case kTrueType: out += "true"; break;
        case kNumberType:
            if (isInt_) {
                snprintf(buf, sizeof buf, "%lld", (long long) int_);
                out += buf;
            } else {
                // Uninitialised reference fields (e.g. ControllerInfo::gap) can be NaN; keep the dump
                // loadable (python json and this shim both read NaN / Infinity / -Infinity).
                if (num_ != num_) { out += "NaN"; break; }
                if (num_ > 1.7976931348623157e308) { out += "Infinity"; break; }
                if (num_ < -1.7976931348623157e308) { out += "-Infinity"; break; }
                // real rapidjson's Writer prints Grisu2's digits: near-shortest, and exact for a correctly rounding reader
                // — NOT for the library's own default reader (number_reader.h), so the reference's dump -> load_from_file
                // can move a value by an ulp.  Modelled by the shortest of %.15g / %.16g / %.17g that strtod reads back.
                for (int prec = 15; prec <= 17; ++prec) {
                    snprintf(buf, sizeof buf, "%.*g", prec, num_);
                    if (strtod(buf, nullptr) == num_) break;
                }
                out += buf;
                if (!strpbrk(buf, ".eEn")) out += ".0";  // keep it a "double" literal
            }
            break;
        case kStringType: {
            out += '"';
            for (char c : str_) {
                switch (c) {
                    case '"': out += "\\\""; break;
                    case '\\': out += "\\\\"; break;
                    case '\n': out += "\\n"; break;
                    case '\t': out += "\\t"; break;
                    case '\r': out += "\\r"; break;
                    default: out += c;
                }
            }
            out += '"';
            break;
        }
        case kArrayType: {
            out += '[';
            for (size_t i = 0; i < arr_.size(); ++i) {
                if (i) out += ',';
                arr_[i].writeTo(out);
            }
            out += ']';
            break;
        }
        case kObjectType: {
            out += '{';
            bool first = true;
            for (ConstMemberIterator it = MemberBegin(); it != MemberEnd(); ++it) {
                if (!first) out += ',';
                first = false;
                it->name.writeTo(out);
                out += ':';
                it->value.writeTo(out);
            }
            out += '}';
            break;
        }
    }
}

// Recursive-descent parser over an in-memory buffer.
struct Parser {
    const char *p, *end;
    bool ok = true;
    size_t line = 1;
    // children of the previous container at each nesting depth: siblings are mostly of one shape, the next one reserves that
    std::vector<unsigned> lastCount;
    size_t depth = 0;
    unsigned &shape() {
        if (lastCount.size() <= depth) lastCount.resize(depth + 1, 0);
        return lastCount[depth];
    }

    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) {
            if (*p == '\n') ++line;
            ++p;
        }
    }
    bool lit(const char *s) {
        size_t n = strlen(s);
        if ((size_t)(end - p) >= n && memcmp(p, s, n) == 0) {
            p += n;
            return true;
        }
        return false;
    }
    void parseString(std::string &out) {
        ++p;  // opening quote
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        unsigned cp = 0;
                        for (int i = 0; i < 4 && p + 1 < end; ++i) {
                            ++p;
                            cp = cp * 16 + (unsigned) (isdigit((unsigned char) *p) ? *p - '0' : (tolower(*p) - 'a' + 10));
                        }
                        if (cp < 0x80) out += (char) cp;
                        else if (cp < 0x800) {
                            out += (char) (0xC0 | (cp >> 6));
                            out += (char) (0x80 | (cp & 0x3F));
                        } else {
                            out += (char) (0xE0 | (cp >> 12));
                            out += (char) (0x80 | ((cp >> 6) & 0x3F));
                            out += (char) (0x80 | (cp & 0x3F));
                        }
                        break;
                    }
                    default: out += *p;
                }
                ++p;
            } else {
                const char *run = p;  // (plain characters are appended a run at a time)
                while (p < end && *p != '"' && *p != '\\') ++p;
                out.append(run, (size_t) (p - run));
            }
        }
        if (p >= end) {
            ok = false;
            return;
        }
        ++p;  // closing quote
    }
    void parseValue(Value &v) {
        ws();
        if (p >= end) {
            ok = false;
            return;
        }
        char c = *p;
        if (c == '{') {
            ++p;
            v.kind_ = kObjectType;
            v.obj_ = new std::vector<Member>();
            ws();
            if (p < end && *p == '}') {
                ++p;
                return;
            }
            v.obj_->reserve(shape());
            ++depth;
            while (ok) {
                ws();
                if (p >= end || *p != '"') {
                    ok = false;
                    return;
                }
                v.obj_->emplace_back();
                Member &m = v.obj_->back();
                m.name.kind_ = kStringType;
                parseString(m.name.str_);
                ws();
                if (p >= end || *p != ':') {
                    ok = false;
                    return;
                }
                ++p;
                parseValue(m.value);
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == '}') {
                    ++p;
                    --depth;
                    shape() = (unsigned) v.obj_->size();
                    return;
                }
                ok = false;
            }
        } else if (c == '[') {
            ++p;
            v.kind_ = kArrayType;
            ws();
            if (p < end && *p == ']') {
                ++p;
                return;
            }
            v.arr_.reserve(shape());
            ++depth;
            while (ok) {
                v.arr_.emplace_back();
                parseValue(v.arr_.back());
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == ']') {
                    ++p;
                    --depth;
                    shape() = (unsigned) v.arr_.size();
                    return;
                }
                ok = false;
            }
        } else if (c == '"') {
            v.kind_ = kStringType;
            parseString(v.str_);
        } else if (c == 't') {
            if (lit("true")) v.kind_ = kTrueType; else ok = false;
        } else if (c == 'f') {
            if (lit("false")) v.kind_ = kFalseType; else ok = false;
        } else if (c == 'n') {
            if (lit("null")) v.kind_ = kNullType; else ok = false;
        } else if (c == 'N' || c == 'I' || (c == '-' && p + 1 < end && p[1] == 'I')) {
            v.kind_ = kNumberType;
            v.isInt_ = false;
            if (lit("NaN")) v.num_ = strtod("nan", nullptr);
            else if (lit("Infinity")) v.num_ = strtod("inf", nullptr);
            else if (lit("-Infinity")) v.num_ = -strtod("inf", nullptr);
            else ok = false;
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            // real rapidjson's default number reader, not strtod (number_reader.h)
            const shim_number::Parsed num = shim_number::read(p, (size_t) (end - p));
            if (!num.ok) {
                ok = false;
                return;
            }
            p += num.length;
            v.kind_ = kNumberType;
            v.num_ = num.value;
            const bool fits = num.isInteger && (num.negative ? num.absInt <= (1ull << 63) : num.absInt < (1ull << 63));
            v.isInt_ = fits;
            v.int_ = fits ? (num.negative ? (int64_t) (0 - num.absInt) : (int64_t) num.absInt) : 0;
        } else {
            ok = false;
        }
    }
};

class Document : public Value {
public:
    typedef MemoryPoolAllocator<> AllocatorType;

    Document() : err_(kParseErrorNone) {}

    template <typename Stream> Document &ParseStream(Stream &is) {
        std::string text;
        is.slurp(text);
        Parser ps;
        ps.p = text.data();
        ps.end = text.data() + text.size();
        clear();
        ps.parseValue(*this);
        ps.ws();
        if (!ps.ok || ps.p != ps.end) {
            err_ = kParseErrorSyntax;
            is.setLine(ps.line);
        } else {
            err_ = kParseErrorNone;
        }
        return *this;
    }
    bool HasParseError() const { return err_ != kParseErrorNone; }
    ParseErrorCode GetParseError() const { return err_; }
    AllocatorType &GetAllocator() { return alloc_; }

    template <typename W> bool Accept(W &writer) const {
        std::string out;
        writeTo(out);
        writer.emit(out);
        return true;
    }

private:
    ParseErrorCode err_;
    AllocatorType alloc_;
};

}  // namespace rapidjson

#endif
