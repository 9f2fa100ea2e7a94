// TEST INFRASTRUCTURE ONLY.  Compares the two restatements of rapidjson's default number reader — the shim's
// (rapidjson_shim/rapidjson/number_reader.h, which the reference build in oracle/_ref reads its files through) and the
// product's (cityflow_amd/csrc/host/json_number.h) — bit for bit on random literals of every shape, and counts how often that
// value differs from the correctly rounded strtod (what both used before round 4).
// usage: probe_json_number [count] [seed]     exit code 1 on any disagreement
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "../cityflow_amd/csrc/host/json_number.h"
#include "rapidjson_shim/rapidjson/number_reader.h"

static uint64_t bits(double d) {
    uint64_t u;
    memcpy(&u, &d, 8);
    return u;
}

int main(int argc, char **argv) {
    const long count = argc > 1 ? atol(argv[1]) : 2000000;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 12345);
    auto digits = [&](int n, bool noLeadingZero) {
        std::string s;
        for (int i = 0; i < n; ++i) s += (char) ('0' + rng() % 10);
        if (noLeadingZero && s[0] == '0') s[0] = '1' + rng() % 9;
        return s;
    };
    long disagree = 0, offStrtod = 0, offStrtodRepr = 0, reprCount = 0, invalid = 0;
    for (long n = 0; n < count; ++n) {
        std::string lit;
        const int shape = (int) (rng() % 8);
        if (shape == 7) {  // what Python's repr / json.dump writes for a random double (the generator's coordinates, archives)
            double x = std::ldexp((double) (rng() >> 11), (int) (rng() % 80) - 90);
            if (rng() & 1) x = -x;
            char buf[40];
            snprintf(buf, sizeof buf, "%.17g", x);
            for (int p = 1; p < 17; ++p) {  // shortest representation that reads back
                char b2[40];
                snprintf(b2, sizeof b2, "%.*g", p, x);
                if (strtod(b2, nullptr) == x) {
                    strcpy(buf, b2);
                    break;
                }
            }
            lit = buf;
            ++reprCount;
        } else {
            if (rng() % 2) lit += '-';
            const int ni = shape == 0 ? 1 + (int) (rng() % 25) : 1 + (int) (rng() % 6);
            lit += rng() % 8 == 0 ? std::string("0") : digits(ni, true);
            if (shape != 0 || rng() % 2) {
                if (rng() % 8) lit += "." + digits(1 + (int) (rng() % (shape == 1 ? 30 : 18)), false);
                if (rng() % 3 == 0) {
                    lit += rng() % 2 ? "e" : "E";
                    if (rng() % 3 == 0) lit += "+";
                    else if (rng() % 2) lit += "-";
                    lit += std::to_string(rng() % (shape == 2 ? 340 : 30));
                }
            }
        }
        const std::string padded = lit + ",";
        const cfa::JsonNumber a = cfa::parseJsonNumber(padded.data(), padded.data() + padded.size());
        const rapidjson::shim_number::Parsed b = rapidjson::shim_number::read(padded.data(), padded.size());
        const bool same = a.ok == b.ok && (!a.ok || (a.integral == b.isInteger && bits(a.d) == bits(b.value) &&
                                                     (size_t) (a.end - padded.data()) == b.length &&
                                                     (!a.integral || (a.magnitude == b.absInt && a.negative == b.negative))));
        if (!same) {
            if (++disagree <= 10) printf("DISAGREE %s: product ok=%d %.17g  shim ok=%d %.17g\n", lit.c_str(), a.ok, a.d, b.ok, b.value);
            continue;
        }
        if (!a.ok) {
            ++invalid;
            continue;
        }
        const double ref = strtod(lit.c_str(), nullptr);
        if (bits(ref) != bits(a.d) && !(ref == 0.0 && a.d == 0.0)) {
            ++offStrtod;
            if (shape == 7) ++offStrtodRepr;
        }
    }
    printf("literals %ld  disagreements %ld  rejected-by-both %ld  differ-from-strtod %ld  (of %ld repr-style: %ld)\n", count, disagree,
           invalid, offStrtod, reprCount, offStrtodRepr);
    return disagree ? 1 : 0;
}
