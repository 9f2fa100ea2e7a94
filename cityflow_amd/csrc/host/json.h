// Minimal JSON DOM for the host loader (config / roadnet / flow files).
//
// Numbers are converted the way the reference's reader converts them — rapidjson's default (not correctly rounded) number
// parser, restated in json_number.h (the library itself is an empty submodule in the reference tree); integer literals keep
// an exact int64.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "json_number.h"

namespace cfa {

struct JsonError : std::runtime_error {
    explicit JsonError(const std::string &m) : std::runtime_error(m) {}
};

class Json {
public:
    enum Kind : uint8_t { Null, False, True, Number, String, Array, Object };

    Kind kind = Null;
    bool integral = false;
    int64_t i = 0;
    double d = 0;
    std::string s;
    std::vector<Json> items;                             // Array
    std::vector<std::pair<std::string, Json>> members;   // Object (file order)
    mutable size_t hint_ = 0;                            // where find() looks first

    bool isObject() const { return kind == Object; }
    bool isArray() const { return kind == Array; }
    bool isString() const { return kind == String; }
    bool isNumber() const { return kind == Number; }
    bool isBool() const { return kind == True || kind == False; }
    bool isInt() const { return kind == Number && integral && i >= INT32_MIN && i <= INT32_MAX; }

    // (the search starts behind the member found last: a reader that asks for an object's members in file order — an Archive's
    // 481 k drivables on a 100x100 grid — pays one comparison per lookup instead of half the object)
    const Json *find(const char *name) const {
        const size_t n = members.size();
        size_t at = hint_ < n ? hint_ : 0;
        for (size_t k = 0; k < n; ++k) {
            if (members[at].first == name) {
                hint_ = at + 1;
                return &members[at].second;
            }
            if (++at == n) at = 0;
        }
        return nullptr;
    }
    // Mirrors the reference's getJsonMember<T> error wording (utility.h:92-136) so config errors read alike.
    const Json &at(const char *name) const {
        const Json *v = find(name);
        if (!v) throw JsonError(std::string(name) + " is required but missing in json file");
        return *v;
    }
    double numberAt(const char *name) const {
        const Json &v = at(name);
        if (!v.isNumber()) throw JsonError(std::string(name) + ": expected type double");
        return v.asDouble();
    }
    int intAt(const char *name) const {
        const Json &v = at(name);
        if (!v.isInt()) throw JsonError(std::string(name) + ": expected type int");
        return (int) v.i;
    }
    int intAt(const char *name, int dflt) const {
        const Json *v = find(name);
        return (v && v->isInt()) ? (int) v->i : dflt;
    }
    bool boolAt(const char *name) const {
        const Json &v = at(name);
        if (!v.isBool()) throw JsonError(std::string(name) + ": expected type bool");
        return v.kind == True;
    }
    bool boolAt(const char *name, bool dflt) const {
        const Json *v = find(name);
        return (v && v->isBool()) ? v->kind == True : dflt;
    }
    const std::string &stringAt(const char *name) const {
        const Json &v = at(name);
        if (!v.isString()) throw JsonError(std::string(name) + ": expected type string");
        return v.s;
    }
    const Json &arrayAt(const char *name) const {
        const Json &v = at(name);
        if (!v.isArray()) throw JsonError(std::string(name) + ": expected type array");
        return v;
    }
    const Json &objectAt(const char *name) const {
        const Json &v = at(name);
        if (!v.isObject()) throw JsonError(std::string(name) + ": expected type object");
        return v;
    }
    double asDouble() const { return integral ? (double) i : d; }

    static Json parseFile(const std::string &path) {
        FILE *fp = fopen(path.c_str(), "rb");
        if (!fp) throw JsonError("cannot open " + path);
        std::string text;
        if (fseek(fp, 0, SEEK_END) == 0) {  // (one allocation of the file's size)
            const long size = ftell(fp);
            if (size > 0) text.reserve((size_t) size);
            fseek(fp, 0, SEEK_SET);
        }
        char buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
        fclose(fp);
        return parseText(text);
    }

    // A file whose top level is an object with a few HUGE members (an Archive of a million vehicles: 1.1 GB, of which the
    // `vehicles` array and the `drivables` object are all but a few KB): the members named in `streamed` are not materialised.
    // Each of their children is parsed into a Json of its own, handed to `child(member, key, value)` — key = the child's name
    // inside an object, empty inside an array — and dropped, so the reader's memory is one child, not the file's DOM (6 GB for
    // that Archive, and tens of seconds of page faults where those are slow).  The returned object holds every other member;
    // a streamed member is left in it as an empty container of its kind.
    template <class F>
    static Json parseFileStreamed(const std::string &path, std::initializer_list<const char *> streamed, F &&child) {
        FILE *fp = fopen(path.c_str(), "rb");
        if (!fp) throw JsonError("cannot open " + path);
        std::string text;
        if (fseek(fp, 0, SEEK_END) == 0) {
            const long size = ftell(fp);
            if (size > 0) text.reserve((size_t) size);
            fseek(fp, 0, SEEK_SET);
        }
        char buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
        fclose(fp);
        Cursor c{text.data(), text.data() + text.size(), 1, {}, 0};
        Json root;
        c.ws();
        if (c.p >= c.end || *c.p != '{') c.fail("expected an object at the top level");
        ++c.p;
        root.kind = Object;
        c.ws();
        if (c.p < c.end && *c.p == '}') return root;
        c.depth = 1;
        for (;;) {
            c.ws();
            if (c.p >= c.end || *c.p != '"') c.fail("expected member name");
            root.members.emplace_back();
            std::string &name = root.members.back().first;
            Json &value = root.members.back().second;
            c.str(name);
            c.ws();
            if (c.p >= c.end || *c.p != ':') c.fail("expected ':'");
            ++c.p;
            c.ws();
            bool stream = false;
            for (const char *s : streamed) stream = stream || name == s;
            if (stream && c.p < c.end && (*c.p == '[' || *c.p == '{')) {
                const bool isObject = *c.p == '{';
                const char close = isObject ? '}' : ']';
                ++c.p;
                value.kind = isObject ? Object : Array;
                c.ws();
                if (c.p < c.end && *c.p == close) {
                    ++c.p;
                } else {
                    c.depth = 2;
                    std::string key;
                    for (;;) {
                        key.clear();
                        if (isObject) {
                            c.ws();
                            if (c.p >= c.end || *c.p != '"') c.fail("expected member name");
                            c.str(key);
                            c.ws();
                            if (c.p >= c.end || *c.p != ':') c.fail("expected ':'");
                            ++c.p;
                        }
                        Json element;
                        c.value(element);
                        child(name, key, element);
                        c.ws();
                        if (c.p < c.end && *c.p == ',') {
                            ++c.p;
                            continue;
                        }
                        if (c.p < c.end && *c.p == close) {
                            ++c.p;
                            break;
                        }
                        c.fail(isObject ? "expected ',' or '}'" : "expected ',' or ']'");
                    }
                    c.depth = 1;
                }
            } else {
                c.value(value);
            }
            c.ws();
            if (c.p < c.end && *c.p == ',') {
                ++c.p;
                continue;
            }
            if (c.p < c.end && *c.p == '}') {
                ++c.p;
                break;
            }
            c.fail("expected ',' or '}'");
        }
        c.ws();
        if (c.p != c.end) c.fail("trailing characters");
        return root;
    }

    static Json parseText(const std::string &text) {
        Cursor c{text.data(), text.data() + text.size(), 1, {}, 0};
        Json root;
        c.value(root);
        c.ws();
        if (c.p != c.end) c.fail("trailing characters");
        return root;
    }

private:
    struct Cursor {
        const char *p, *end;
        size_t line;
        // how many children the previous container at each nesting depth had: siblings are mostly of one shape (an Archive's
        // vehicles: 97 k objects of 35 members), so the next one reserves exactly that and its vector never reallocates
        std::vector<uint32_t> lastCount;
        size_t depth;
        uint32_t &shape() {
            if (lastCount.size() <= depth) lastCount.resize(depth + 1, 0);
            return lastCount[depth];
        }

        [[noreturn]] void fail(const char *what) const {
            throw JsonError("Json parsing error at line " + std::to_string(line) + ": " + what);
        }
        void ws() {
            while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) {
                if (*p == '\n') ++line;
                ++p;
            }
        }
        void expect(const char *lit) {
            size_t n = strlen(lit);
            if ((size_t) (end - p) < n || memcmp(p, lit, n) != 0) fail("invalid literal");
            p += n;
        }
        void str(std::string &out) {
            ++p;
            for (;;) {
                const char *run = p;  // (plain characters are appended a run at a time)
                while (p < end && *p != '"' && *p != '\\') ++p;
                if (p > run) out.append(run, (size_t) (p - run));
                if (p >= end) fail("unterminated string");
                char ch = *p++;
                if (ch == '"') return;
                if (p >= end) fail("bad escape");
                char e = *p++;
                switch (e) {
                    case 'n': out.push_back('\n'); break;
                    case 't': out.push_back('\t'); break;
                    case 'r': out.push_back('\r'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'u': {
                        if (end - p < 4) fail("bad \\u escape");
                        unsigned cp = (unsigned) strtoul(std::string(p, 4).c_str(), nullptr, 16);
                        p += 4;
                        if (cp < 0x80) out.push_back((char) cp);
                        else if (cp < 0x800) {
                            out.push_back((char) (0xC0 | (cp >> 6)));
                            out.push_back((char) (0x80 | (cp & 0x3F)));
                        } else {
                            out.push_back((char) (0xE0 | (cp >> 12)));
                            out.push_back((char) (0x80 | ((cp >> 6) & 0x3F)));
                            out.push_back((char) (0x80 | (cp & 0x3F)));
                        }
                        break;
                    }
                    default: out.push_back(e);
                }
            }
        }
        void value(Json &v) {
            ws();
            if (p >= end) fail("unexpected end of input");
            switch (*p) {
                case '{': {
                    ++p;
                    v.kind = Object;
                    ws();
                    if (p < end && *p == '}') {
                        ++p;
                        return;
                    }
                    v.members.reserve(shape());
                    ++depth;
                    for (;;) {
                        ws();
                        if (p >= end || *p != '"') fail("expected member name");
                        v.members.emplace_back();
                        str(v.members.back().first);
                        ws();
                        if (p >= end || *p != ':') fail("expected ':'");
                        ++p;
                        value(v.members.back().second);
                        ws();
                        if (p < end && *p == ',') {
                            ++p;
                            continue;
                        }
                        if (p < end && *p == '}') {
                            ++p;
                            --depth;
                            shape() = (uint32_t) v.members.size();
                            return;
                        }
                        fail("expected ',' or '}'");
                    }
                }
                case '[': {
                    ++p;
                    v.kind = Array;
                    ws();
                    if (p < end && *p == ']') {
                        ++p;
                        return;
                    }
                    v.items.reserve(shape());
                    ++depth;
                    for (;;) {
                        v.items.emplace_back();
                        value(v.items.back());
                        ws();
                        if (p < end && *p == ',') {
                            ++p;
                            continue;
                        }
                        if (p < end && *p == ']') {
                            ++p;
                            --depth;
                            shape() = (uint32_t) v.items.size();
                            return;
                        }
                        fail("expected ',' or ']'");
                    }
                }
                case '"':
                    v.kind = String;
                    str(v.s);
                    return;
                case 't':
                    expect("true");
                    v.kind = True;
                    return;
                case 'f':
                    expect("false");
                    v.kind = False;
                    return;
                case 'n':
                    expect("null");
                    v.kind = Null;
                    return;
                case 'N':  // NaN / Infinity are not JSON, but the reference dumps uninitialised doubles
                    expect("NaN");  // (ControllerInfo::gap of vehicles without a leader, archive.cpp:218)
                    v.kind = Number;
                    v.d = std::strtod("nan", nullptr);
                    return;
                case 'I':
                    expect("Infinity");
                    v.kind = Number;
                    v.d = std::strtod("inf", nullptr);
                    return;
                default: {
                    if (end - p >= 9 && memcmp(p, "-Infinity", 9) == 0) {
                        p += 9;
                        v.kind = Number;
                        v.d = -std::strtod("inf", nullptr);
                        return;
                    }
                    // the reference's reader (rapidjson, default flags), not strtod: json_number.h
                    const JsonNumber num = parseJsonNumber(p, end);
                    if (!num.ok) fail("invalid value");
                    p = num.end;
                    v.kind = Number;
                    v.d = num.d;
                    // Int / Uint / Int64 / Uint64 events; beyond the int64 range the value is kept as its double only
                    if (num.integral && (num.negative ? num.magnitude <= (1ull << 63) : num.magnitude < (1ull << 63))) {
                        v.integral = true;
                        v.i = num.negative ? (int64_t) (~num.magnitude + 1) : (int64_t) num.magnitude;
                    }
                    return;
                }
            }
        }
    };
};

}  // namespace cfa
