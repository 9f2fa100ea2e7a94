"""CPU: decimal literal -> double exactly as the reference's JSON reader does it.

The reference reads every file through rapidjson with the DEFAULT parse flags (reference src/utility/utility.cpp:96-114), whose
number reader is not a correctly rounded strtod (GenericReader::ParseNumber + internal::StrtodNormalPrecision).  rapidjson is an
empty submodule of the reference tree, so its published algorithm is restated three times, independently written, and the three
are compared here:
  * csrc/host/json_number.h            the product's host loader (over the character stream),
  * oracle/rapidjson_shim/.../number_reader.h   what the reference build of oracle/_ref reads through (over digit groups),
  * `reader` below                     plain Python integers and floats.
What stays unpinned is the library itself: it is not here to run."""
import os
import random
import struct
import subprocess

import pytest

from conftest import REF_DIR


def reader(lit):
    """rapidjson's ParseNumber (64-bit build, no kParseFullPrecisionFlag) -> (is_integer_event, value as GetDouble())."""
    i, n = 0, len(lit)
    minus = lit[0] == "-"
    if minus:
        i = 1
    digit = lambda k: k < n and lit[k].isdigit()
    sig, d, in_double, counted = 0, 0.0, False, 0
    if lit[i] == "0":
        i += 1
    else:
        sig = int(lit[i])
        i += 1
        lim32, last32 = (214748364, "8") if minus else (429496729, "5")
        wide = False
        while digit(i):
            if sig >= lim32 and (sig != lim32 or lit[i] > last32):
                wide = True
                break
            sig = sig * 10 + int(lit[i])
            i += 1
            counted += 1
        if wide:
            lim64, last64 = (0x0CCCCCCCCCCCCCCC, "8") if minus else (0x1999999999999999, "5")
            while digit(i):
                if sig >= lim64 and (sig != lim64 or lit[i] > last64):
                    d, in_double = float(sig), True
                    break
                sig = sig * 10 + int(lit[i])
                i += 1
                counted += 1
        if in_double:
            while digit(i):
                d = d * 10 + int(lit[i])
                i += 1
    real, taken = in_double, 0
    if i < n and lit[i] == ".":
        i += 1
        if not in_double:
            while digit(i):
                if sig > 0x1FFFFFFFFFFFFF:
                    break
                sig = sig * 10 + int(lit[i])
                i += 1
                taken += 1
                if sig:
                    counted += 1
            d, in_double = float(sig), True
        while digit(i):
            if counted < 17:
                d = d * 10.0 + int(lit[i])
                taken += 1
                if d > 0.0:
                    counted += 1
            i += 1
        real = True
    exp = 0
    if i < n and lit[i] in "eE":
        i += 1
        if not in_double:
            d, in_double = float(sig), True
        real = True
        neg = False
        if lit[i] in "+-":
            neg = lit[i] == "-"
            i += 1
        exp = int(lit[i])
        i += 1
        if neg:
            max_exp = (-taken + 2147483639) // 10
            while digit(i):
                exp = exp * 10 + int(lit[i])
                i += 1
                if exp > max_exp:
                    while digit(i):
                        i += 1
            exp = -exp
        else:
            while digit(i):
                exp = exp * 10 + int(lit[i])
                i += 1
    if not real:
        return True, float(-sig if minus else sig)
    p = exp - taken
    once = lambda x, q: 0.0 if q < -308 else (x * float("1e%d" % q) if q >= 0 else x / float("1e%d" % -q))
    d = once(once(d, -308), p + 308) if p < -308 else once(d, p)
    return False, -d if minus else d


def random_literal(rng):
    shape = rng.randrange(7)
    if shape == 6:  # what Python's json writes for a double
        x = struct.unpack("<d", struct.pack("<Q", rng.getrandbits(52) | (rng.randrange(950, 1100) << 52)))[0]
        return repr(-x if rng.random() < 0.5 else x)
    s = "-" if rng.random() < 0.5 else ""
    s += "0" if rng.random() < 0.12 else str(rng.randrange(1, 10)) + "".join(rng.choice("0123456789") for _ in range(rng.randrange(24 if shape == 0 else 6)))
    if shape != 0 or rng.random() < 0.5:
        if rng.random() < 0.9:
            s += "." + "".join(rng.choice("0123456789") for _ in range(rng.randrange(1, 30 if shape == 1 else 19)))
        if rng.random() < 0.3:
            s += rng.choice("eE") + rng.choice(["", "+", "-"]) + str(rng.randrange(300 if shape == 2 else 30))
    return s


def test_product_reader_equals_the_python_restatement(mod):
    rng = random.Random(20240924)
    off = 0
    for _ in range(60000):
        lit = random_literal(rng)
        is_int, want = reader(lit)
        if want == float("inf") or want == float("-inf"):
            continue  # rapidjson: kParseErrorNumberTooBig
        got, got_int = mod._parse_json_number(lit)
        assert struct.pack("<d", got) == struct.pack("<d", want), lit
        assert got_int == (is_int and abs(want) < 2.0 ** 63), lit  # (an Uint64 beyond the int64 range is kept as its double here)
        off += float(lit) != want
    assert off > 1000  # (it really is a different function from strtod)


def test_shim_reader_equals_product_reader():
    probe = os.path.join(REF_DIR, "probe_json_number")
    if not os.path.exists(probe):
        pytest.skip("oracle/_ref/probe_json_number not built")
    out = subprocess.run([probe, "400000", "77"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert " disagreements 0 " in out.stdout


def test_short_literals_are_read_like_strtod(mod):
    """Up to 15 significant digits and a small exponent: one exact integer, one exact power of ten, one rounding."""
    rng = random.Random(5)
    for lit in ["0", "-0", "0.0", "-0.0", "1", "16.67", "0.1", "2.5", "300", "1e-3", "4.5", "11.111", "1E5", "123456789012345"]:
        assert mod._parse_json_number(lit)[0] == float(lit), lit
    for _ in range(20000):
        lit = "%.*g" % (rng.randrange(1, 16), rng.uniform(-1, 1) * 10 ** rng.randrange(-6, 7))
        assert mod._parse_json_number(lit)[0] == float(lit), lit
    assert mod._parse_json_number("-0")[0] == 0.0 and str(mod._parse_json_number("-0")[0]) == "0.0"  # the INTEGER zero
    assert str(mod._parse_json_number("-0.0")[0]) == "-0.0"
    assert mod._parse_json_number("18446744073709551616")[0] == 2.0 ** 64  # beyond the 64-bit accumulator


def test_archive_writer_survives_both_readers(mod):
    """Archive.dump's literal for a double: read back exactly by the reference's reader (what load_from_file of either engine
    uses) — and, wherever such a literal exists, by a correctly rounding reader too."""
    rng = random.Random(9)
    n, lost, strtod_off = 150000, 0, 0
    for i in range(n):
        v = rng.uniform(0, 400) if i % 2 else rng.uniform(-1e4, 1e4) * 10 ** rng.randrange(-6, 2)
        lit = mod._format_json_number(v)
        back = mod._parse_json_number(lit)[0]
        assert back == reader(lit)[1]
        lost += back != v
        strtod_off += float(lit) != v
    assert lost <= n * 3e-5, lost            # (doubles no literal reaches through that reader: about 1 in 10^5)
    assert strtod_off <= n * 5e-4, strtod_off
    for v in (0.0, 1.0, -1.0, 0.5, 16.67, 1e22, 1e-7, 123456.789):
        assert float(mod._format_json_number(v)) == v
    assert mod._format_json_number(float("nan")) == "NaN" and mod._format_json_number(float("inf")) == "Infinity"


def test_reader_meets_rapidjsons_own_unit_test_expectations(mod):
    """The literals rapidjson's own reader tests run through BOTH readers (test/unittest/readertest.cpp, `TestParseDouble<false>`:
    the default, not-full-precision one is held to EXPECT_DOUBLE_EQ, i.e. within 4 units in the last place of the exact value;
    the underflowing ones to exactly 0).  Not a pin of the bits — the library publishes none for this reader — but what it
    promises of it."""
    def ulps(a, b):
        ia, ib = struct.unpack("<q", struct.pack("<d", a))[0], struct.unpack("<q", struct.pack("<d", b))[0]
        return abs(ia - ib)

    for lit in ["0.0", "-0.0", "1.0", "-1.0", "1.5", "-1.5", "3.1416", "1E10", "1e10", "1E+10", "1E-10", "-1E10", "-1e10", "-1E+10",
                "-1E-10", "1.234E+10", "1.234E-10", "1.79769e+308", "2.22507e-308", "-1.79769e+308", "-2.22507e-308",
                "4.9406564584124654e-324", "2.2250738585072009e-308", "2.2250738585072014e-308", "1.7976931348623157e+308",
                "18446744073709551616", "-9223372036854775809", "0.9868011474609375", "123e34", "45913141877270640000.0",
                "2.2250738585072011e-308", "0.017976931348623157e+310", "2.2250738585072012e-308", "0.999999999999999944488848768742172978818416595458984375",
                "1.00000000000000011102230246251565404236316680908203125", "72057594037927928.0", "72057594037927936.0",
                "9223372036854774784.0", "9223372036854775808.0", "10141204801825834086073718800384", "5708990770823839207320493820740630171355185151999e-3",
                "2.225073858507201136057409796709131975934819546351645648023426109724822222021076945516529523908135087914149158913039621106870086438694594645527657207407820621743379988141063267329253552286881372149012981122451451889849057222307285255133155755015914397476397983411801999323962548289017107081850690630666655994938275772572015763062690663332647565300009245888316433037779791869612049497390377829704905051080609940730262937128958950003583799967207254304360284078895771796150945516748243471030702609144621572289880258182545180325707018860872113128079512233426288368622321503775666622503982534335974568884423900265498198385487948292206894721689831099698365846814022854243330660339850886445804001034933970427567186443383770486037861622771738545623065874679014086723327636718749999999999999999999999999999999999999e-308"]:
        got = mod._parse_json_number(lit)[0]
        want = float(lit)
        assert ulps(got, want) <= 4, (lit, got, want)
        assert got == reader(lit)[1]
    for lit in ["1e-10000", "1e-00011111111111", "-1e-00011111111111", "1e-214748363", "1e-214748364", "1e-21474836311"]:
        got = mod._parse_json_number(lit)[0]
        assert got == 0.0 and (str(got) == "-0.0") == lit.startswith("-"), lit


def test_archive_literals_are_checked_as_emitted_and_the_rest_is_counted():
    """Archive.dump's number writer (csrc/host/archive.cpp num()): every literal is verified in the form it is written in —
    the ".0" / "e0" an integer-valued double gets included (round 4 checked the bare digits and appended ".0" afterwards:
    4 % of the integer doubles in [2^53, 1e17) then read back wrong through the reference's fraction path) — and the doubles
    for which NO literal makes the reference's reader return them (mantissa close to 2, about one in 10^5 at simulation
    magnitudes; the reference's own dumps come back an ulp off there too) are written with 17 digits and COUNTED, never
    silently: Archive.dump reports the count (Archive._last_dump_inexact, a warning on stderr)."""
    import random
    from cityflow_amd import _cityflow as m
    fmt, parse, counted = m._format_json_number, m._parse_json_number, m._inexact_json_numbers
    import math
    rng = random.Random(20250925)
    n, off, only_ref = 120000, 0, 0
    before = counted()
    for _ in range(n):
        x = rng.uniform(0.0, 3000.0)
        lit = fmt(x)
        ref_exact, strtod_exact = parse(lit)[0] == x, float(lit) == x
        assert ref_exact or strtod_exact, (x, lit)  # at least one kind of reader always gets the value back
        assert abs(float(lit) - x) <= 4 * math.ulp(x) and abs(parse(lit)[0] - x) <= math.ulp(x), (x, lit)
        if not strtod_exact:
            only_ref += 1                           # (a literal for the reference's reader only: the file is for engines to load)
        if not ref_exact:
            off += 1
    assert off == counted() - before                # every value the reference's reader misses was counted
    assert off <= n // 5000, off                    # ... and they are rare (measured: ~1 in 10^5)
    assert only_ref <= n // 500, only_ref           # (measured: ~1 in 10^4)
    before = counted()
    for _ in range(40000):
        x = float(rng.randrange(2 ** 53, 10 ** 17))
        lit = fmt(x)
        assert float(lit) == x and parse(lit)[0] == x, (x, lit)
        assert any(c in lit for c in ".eE"), lit    # (never an integer event for a field the loader reads as a double)
    assert counted() == before
    for x in (0.0, 1.0, 5.0, 16.67, 1e-300, 123456789012345680.0, 2.0 ** 53, 2.0 ** 63, 1e22, 1e23):
        lit = fmt(x)
        assert float(lit) == x and parse(lit)[0] == x, (x, lit)
