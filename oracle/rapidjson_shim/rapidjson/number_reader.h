// TEST INFRASTRUCTURE ONLY (oracle/): how real rapidjson turns a number literal into a value with its DEFAULT parse flags
// (GenericReader::ParseNumber without kParseFullPrecisionFlag, internal::StrtodNormalPrecision / FastPath / Pow10 of
// Tencent/rapidjson >= 1.1.0, 64-bit build) — which is what the reference calls (`document.ParseStream(csw)`,
// reference src/utility/utility.cpp:105).  rapidjson itself is absent from the reference tree (empty submodule); this is the
// shim's own restatement of the published algorithm, written over the literal's digit groups (the product has a second,
// independently written one over the character stream: cityflow_amd/csrc/host/json_number.h; oracle/probe_json_number.cpp
// compares the two and strtod on random literals).
//   value = S x 10^(E - f): S is the significand as rapidjson accumulates it — an unsigned integer while the next digit
//   cannot overflow the 32-bit (then the 64-bit) accumulator, continued in a double (x 10 + digit per further digit);
//   fraction digits keep going into the integer until it exceeds 2^53 - 1, then into the double for as long as fewer than
//   17 significant digits have been taken (the rest is dropped); f = fraction digits actually taken; the power of ten is
//   applied with ONE multiplication or division by the double nearest to 10^k (two below 10^-308).
#ifndef ORACLE_RAPIDJSON_SHIM_NUMBER_READER_H
#define ORACLE_RAPIDJSON_SHIM_NUMBER_READER_H

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>

namespace rapidjson {
namespace shim_number {

struct Parsed {
    bool ok, isInteger, negative;
    uint64_t absInt;
    double value;
    size_t length;  // characters of the literal
};

inline double tenTo(int k) {  // rapidjson's table holds the literals 1e0..1e308, i.e. the correctly rounded powers
    static double table[309];
    static bool ready = false;
    if (!ready) {
        char buf[16];
        for (int i = 0; i <= 308; ++i) {
            snprintf(buf, sizeof buf, "1e%d", i);
            table[i] = strtod(buf, nullptr);
        }
        ready = true;
    }
    return table[k];
}

inline double applyPower(double s, int p) {
    auto once = [](double x, int q) { return q < -308 ? 0.0 : (q >= 0 ? x * tenTo(q) : x / tenTo(-q)); };
    return p < -308 ? once(once(s, -308), p + 308) : once(s, p);
}

inline Parsed read(const char *text, size_t avail) {
    Parsed r{false, false, false, 0, 0.0, 0};
    size_t at = 0;
    auto isDigit = [&](size_t i) { return i < avail && text[i] >= '0' && text[i] <= '9'; };
    if (at < avail && text[at] == '-') {
        r.negative = true;
        ++at;
    }
    // --- the literal's three digit groups
    std::string intPart, fracPart;
    if (!isDigit(at)) return r;
    if (text[at] == '0') intPart = "0", ++at;  // a leading zero stands alone
    else
        while (isDigit(at)) intPart += text[at++];
    bool hasFrac = false, hasExp = false, expNeg = false;
    std::string expPart;
    if (at < avail && text[at] == '.') {
        hasFrac = true;
        ++at;
        if (!isDigit(at)) return r;
        while (isDigit(at)) fracPart += text[at++];
    }
    if (at < avail && (text[at] == 'e' || text[at] == 'E')) {
        hasExp = true;
        ++at;
        if (at < avail && (text[at] == '+' || text[at] == '-')) expNeg = text[at++] == '-';
        if (!isDigit(at)) return r;
        while (isDigit(at)) expPart += text[at++];
    }
    r.length = at;

    // --- the significand
    uint64_t acc = (uint64_t) (intPart[0] - '0');
    double dacc = 0.0;
    bool inDouble = false;
    int counted = 0;  // "significandDigit"
    {
        const uint64_t cap32 = r.negative ? 214748364ull : 429496729ull, cap64 = r.negative ? 0x0CCCCCCCCCCCCCCCull : 0x1999999999999999ull;
        const char edge = r.negative ? '8' : '5';
        size_t i = 1;
        int stage = 32;
        for (; i < intPart.size(); ++i) {
            const char ch = intPart[i];
            const uint64_t cap = stage == 32 ? cap32 : cap64;
            if (acc >= cap && (acc != cap || ch > edge)) {
                if (stage == 32) {
                    stage = 64;
                    --i;  // the same digit is offered to the 64-bit stage
                    continue;
                }
                break;  // does not fit 64 bits either
            }
            acc = acc * 10 + (uint64_t) (ch - '0');
            ++counted;
        }
        if (i < intPart.size()) {
            inDouble = true;
            dacc = (double) acc;
            for (; i < intPart.size(); ++i) dacc = dacc * 10 + (intPart[i] - '0');
        }
    }
    int fracTaken = 0;
    if (hasFrac) {
        size_t i = 0;
        if (!inDouble) {
            for (; i < fracPart.size(); ++i) {
                if (acc > 0x1FFFFFFFFFFFFFull) break;
                acc = acc * 10 + (uint64_t) (fracPart[i] - '0');
                ++fracTaken;
                if (acc != 0) ++counted;
            }
            dacc = (double) acc;
            inDouble = true;
        }
        for (; i < fracPart.size(); ++i) {
            if (counted >= 17) break;  // the remaining digits are read and ignored
            dacc = dacc * 10.0 + (fracPart[i] - '0');
            ++fracTaken;
            if (dacc > 0.0) ++counted;
        }
    }
    if (!hasFrac && !hasExp && !inDouble) {
        r.ok = true;
        r.isInteger = true;
        r.absInt = acc;
        r.value = r.negative ? (double) (int64_t) (0 - acc) : (double) acc;
        return r;
    }
    if (!inDouble) dacc = (double) acc;
    long e = 0;
    if (hasExp) {
        const long bound = expNeg ? (-(long) fracTaken + 2147483639L) / 10 : 308 + fracTaken;
        for (size_t i = 0; i < expPart.size(); ++i) {
            if (i > 0 && expNeg && e > bound) break;  // rapidjson stops reading a hopeless negative exponent
            e = e * 10 + (expPart[i] - '0');
            if (i > 0 && !expNeg && e > bound) return r;  // "number too big"
        }
        if (expNeg) e = -e;
    }
    double v = applyPower(dacc, (int) e - fracTaken);
    if (v > 1.7976931348623157e308) return r;
    r.ok = true;
    r.value = r.negative ? -v : v;
    return r;
}

}  // namespace shim_number
}  // namespace rapidjson
#endif
