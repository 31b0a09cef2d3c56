"""The JSON reader / writer of the product's host library (karpenter_amd/host/json_mini.hpp) and the oracle's (oracle/json_mini.hpp)
are the same source under two namespaces — a parse bug would be common to checker and product and no parity test could see it
(round-4 review). Both are held here against an INDEPENDENT implementation: Python's json. Every document is parsed and written
back by each library (`ksched_json_roundtrip`, `oracle_json_roundtrip`), and what Python reads out of that must equal what Python
reads out of the original."""
import ctypes
import json
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _roundtrippers():
    import oracle
    olib = ctypes.CDLL(oracle.build())
    olib.oracle_json_roundtrip.restype = ctypes.c_void_p
    olib.oracle_json_roundtrip.argtypes = [ctypes.c_char_p]
    olib.oracle_free.argtypes = [ctypes.c_void_p]
    klib = ctypes.CDLL(os.path.join(ROOT, "karpenter_amd", "libksched.so"))
    klib.ksched_json_roundtrip.restype = ctypes.c_void_p
    klib.ksched_json_roundtrip.argtypes = [ctypes.c_char_p]
    klib.ksched_free.argtypes = [ctypes.c_void_p]

    def run(fn, free, doc):
        ptr = fn(doc.encode("utf-8"))
        try:
            return ctypes.string_at(ptr).decode("utf-8")
        finally:
            free(ptr)
    return {"oracle": lambda d: run(olib.oracle_json_roundtrip, olib.oracle_free, d),
            "host library": lambda d: run(klib.ksched_json_roundtrip, klib.ksched_free, d)}


DOCS = [
    '{}', '[]', '[[]]', '{"a":{}}', ' \t\r\n{ "a" : [ 1 , 2 ] , "b" : null }\n',
    '{"t":true,"f":false,"n":null}',
    '[0,-0,1,-1,10,123456789,2147483647,-2147483648,4294967296,9007199254740993,-9223372036854775808,9223372036854775807]',
    '[0.0,-0.0,0.1,-0.25,1.5,3.141592653589793,1e3,1E3,1e+3,1e-3,2.5E-7,1.7976931348623157e308,5e-324,123.456e2]',
    r'["","a","\"","\\","\/","\b\f\n\r\t","\u0041\u00e9\u4e2d","\ud83d\ude00","tab\there","nul\u0000byte","\u001f"]',
    '["é","中文","😀","a/b","cpu=1000m","karpenter.sh/nodepool","topology.kubernetes.io/zone"]',
    '{"":1,"a b":2,"a\\"b":3,"\\u00e9":4,"é2":5,"k/e.y-1_":6}',
    '{"requests":{"cpu":"1500m","memory":"1Gi"},"count":1000000,"uidSeed":4200126,"price":0.000123,"weights":[1,2.0,3e0]}',
    '[' * 60 + ']' * 60,
    '{"a":' * 40 + '1' + '}' * 40,
    json.dumps({"k%d" % i: ["v" * (i % 7), i, i / 7.0, None, i % 2 == 0, {"n": [i] * (i % 5)}] for i in range(200)}),
    json.dumps(["x" * 70000, "é" * 5000]),
]


@pytest.mark.parametrize("name", ["oracle", "host library"])
def test_valid_documents_round_trip_like_python(name):
    rt = _roundtrippers()[name]
    for doc in DOCS:
        want = json.loads(doc)
        got = json.loads(rt(doc))
        assert got == want, (name, doc[:80])   # (values: the writer prints an integral double without its ".0" and -0.0 as 0 — equal numbers; an integer beyond 2**53 that had gone through a double would not compare equal)


@pytest.mark.parametrize("name", ["oracle", "host library"])
def test_generated_documents_round_trip_like_python(name):
    rt = _roundtrippers()[name]
    rng = random.Random(20260923)
    alphabet = ['a', 'Z', '0', ' ', '"', '\\', '/', '\n', '\t', '\x01', 'é', '中', '😀', '{', ']', ':', ',']

    def value(depth):
        k = rng.randrange(8 if depth < 5 else 5)
        if k == 0: return None
        if k == 1: return rng.random() < 0.5
        if k == 2: return rng.choice([0, 1, -1, rng.randrange(-2 ** 62, 2 ** 62), rng.randrange(-1000, 1000)])
        if k == 3: return rng.choice([0.5, -1.25, rng.random(), rng.uniform(-1e12, 1e12), rng.random() * 1e-9, float(rng.randrange(10 ** 15)) * 1e5])
        if k == 4: return "".join(rng.choice(alphabet) for _ in range(rng.randrange(12)))
        if k in (5, 6): return [value(depth + 1) for _ in range(rng.randrange(5))]
        return {"".join(rng.choice(alphabet) for _ in range(rng.randrange(1, 6))) + str(i): value(depth + 1) for i in range(rng.randrange(5))}

    for i in range(300):
        v = value(0)
        doc = json.dumps(v, ensure_ascii=rng.random() < 0.5, indent=rng.choice([None, None, 1]), separators=rng.choice([None, (",", ":")]))
        got = json.loads(rt(doc))
        assert got == v, (name, i, doc[:120])


@pytest.mark.parametrize("name", ["oracle", "host library"])
def test_broken_documents_are_refused(name):
    rt = _roundtrippers()[name]
    for doc in ['', '{', '[1,2', '{"a":}', '{"a" 1}', '["unterminated]', '[1,,2]', '{"a":1,}', 'nul', '[1 2]', '"\\x41"', '{"a":1}}', '[1]]']:
        with pytest.raises(json.JSONDecodeError):
            json.loads(doc)
        out = json.loads(rt(doc))
        assert isinstance(out, dict) and "error" in out, (name, doc, out)


@pytest.mark.parametrize("name", ["oracle", "host library"])
def test_lone_surrogates_are_refused(name):
    """an unpaired \\uD800-\\uDFFF (alone, a high one followed by a non-low escape, a low one first) has no UTF-8 form; both
    parsers used to emit a three-byte sequence for it. Refused now — Python accepts such escapes into a str that cannot be encoded."""
    rt = _roundtrippers()[name]
    for doc in ['"\\ud800"', '"\\udc00"', '"a\\ud83dz"', '"\\ud83d\\u0041"', '"\\ude00\\ud83d"', '["ok", "\\udfff"]']:
        with pytest.raises(UnicodeEncodeError):
            json.loads(doc).__str__().encode("utf-8") if not isinstance(json.loads(doc), list) else json.loads(doc)[1].encode("utf-8")
        out = json.loads(rt(doc))
        assert isinstance(out, dict) and "error" in out, (name, doc, out)
    assert json.loads(rt('"\\ud83d\\ude00"')) == "\U0001F600"      # the pair itself stays one code point
