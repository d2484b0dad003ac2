#!/usr/bin/env python3
"""Extract the reference's NUMERIC DATA TABLES into binary/JSON data files.

Reads (as text, read-only) the data-only parts of the reference:
  BlueNoiseTile.cpp   g_rgvBlueNoise[1600]                 -> product table (sample positions)
                      g_BlueNoiseQuasiIndex{2,3,5,7}[1600] -> GOLDEN vectors (outputs of BuildQuasiIndex)
  LowDiscrepancy.cpp  DiskSamples{16,32,64,128}            -> product tables
                      rgnQuasiIndex{2,3,5,7}[128]          -> GOLDEN vectors
  RadiometryColor.cpp `Spectrum g_sXxx = {...}` constants, SpectrumToRGB[10], g_rgbXxx metal colours
No code is copied: only numbers.  Outputs:
  ray-tracer_amd/data/tables.bin      (product constants; embedded into the host libraries)
  tests/golden/quasi_index_tables.json (golden vectors pinning BuildQuasiIndex)

Run in the build container only (the GPU box has no /root/reference): python tools/extract_reference_tables.py
"""
import json, os, re, struct, sys
import numpy as np

REF = os.environ.get("RT_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def f32_exact(txt):
    """Correctly rounded decimal -> binary32 (what a C compiler does for an `f`-suffixed literal),
    avoiding the double-rounding of float32(float(txt))."""
    from decimal import Decimal
    txt = txt.rstrip("fF")
    d = Decimal(txt)
    c = np.float32(float(txt))
    best = c
    for cand in (np.nextafter(c, np.float32(-np.inf)), np.nextafter(c, np.float32(np.inf))):
        e0 = abs(Decimal(float(best)) - d)
        e1 = abs(Decimal(float(cand)) - d)
        if e1 < e0 or (e1 == e0 and (int(np.float32(cand).view(np.uint32)) & 1) == 0 and (int(np.float32(best).view(np.uint32)) & 1) == 1):
            best = cand
    return np.float32(best)

def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    return re.sub(r"//[^\n]*", "", s)

def read(name):
    with open(os.path.join(REF, name), "r", errors="replace") as f:
        return strip_comments(f.read())

def vec2_table(src, name):
    m = re.search(r"Vector2\s+%s\s*\[\s*\d+\s*\]\s*=\s*\{(.*?)\};" % re.escape(name), src, re.S)
    assert m, name
    vals = re.findall(r"Vector2\(\s*([-+0-9.eE]+)\s*,\s*([-+0-9.eE]+)\s*\)", m.group(1))
    return np.array([[float(a), float(b)] for a, b in vals], dtype=np.float64).reshape(-1)

def int_table(src, name):
    m = re.search(r"int\s+%s\s*\[\s*\d+\s*\]\s*=\s*\{(.*?)\};" % re.escape(name), src, re.S)
    assert m, name
    return np.array([int(x) for x in re.findall(r"-?\d+", m.group(1))], dtype=np.int32)

def main():
    tables = {}   # name -> np array (f32 / f64 / i32)
    golden = {}
    bn = read("BlueNoiseTile.cpp")
    tables["g_rgvBlueNoise"] = vec2_table(bn, "g_rgvBlueNoise")
    assert tables["g_rgvBlueNoise"].size == 3200
    for p in (2, 3, 5, 7):
        t = int_table(bn, "g_BlueNoiseQuasiIndex%d" % p)
        assert t.size == 1600
        golden["BuildQuasiIndex_p%d_N1600" % p] = t.tolist()
    ld = read("LowDiscrepancy.cpp")
    for n in (16, 32, 64, 128):
        t = vec2_table(ld, "DiskSamples%d" % n)
        assert t.size == 2 * n
        tables["DiskSamples%d" % n] = t
    for p in (2, 3, 5, 7):
        t = int_table(ld, "rgnQuasiIndex%d" % p)
        assert t.size == 128
        golden["BuildQuasiIndex_p%d_N128" % p] = t.tolist()
    rc = read("RadiometryColor.cpp")
    for m in re.finditer(r"Spectrum\s+(g_s\w+)\s*=\s*\{(.*?)\};", rc, re.S):
        vals = re.findall(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?f?", m.group(2))
        arr = np.zeros(10, dtype=np.float32)
        assert 0 < len(vals) <= 10, (m.group(1), len(vals))
        for i, v in enumerate(vals):
            arr[i] = f32_exact(v) if v.endswith("f") else np.float32(float(v))
        tables[m.group(1)] = arr
    m = re.search(r"DisplayRGB\s+SpectrumToRGB\s*\[\s*NSPEC\s*\]\s*=\s*\{(.*?)\};", rc, re.S)
    rows = re.findall(r"DisplayRGB\(\s*([-+0-9.eE]+)\s*,\s*([-+0-9.eE]+)\s*,\s*([-+0-9.eE]+)\s*\)", m.group(1))
    assert len(rows) == 10
    # DisplayRGB(double,double,double) rounds each component to float (Image.h:65)
    tables["SpectrumToRGB"] = np.array([[np.float32(float(x)) for x in r] for r in rows], dtype=np.float32).reshape(-1)
    for m in re.finditer(r"(g_rgb\w+)\(\s*([-+0-9.eE]+)\s*,\s*([-+0-9.eE]+)\s*,\s*([-+0-9.eE]+)\s*\)", rc):
        tables[m.group(1)] = np.array([np.float32(float(m.group(i))) for i in (2, 3, 4)], dtype=np.float32)

    # Solids.cpp: the vector font of SolidText, rgFont[ML_NFONT] = up to 8 pen positions per glyph as one byte each
    # (x in the high nibble, y in the low one; 0xFE lifts the pen, 0xFF ends the glyph; short rows are zero-filled by C)
    so = read("Solids.cpp")
    m = re.search(r"VectorFont\s+rgFont\s*\[\s*ML_NFONT\s*\]\s*=\s*\{(.*?)\n\};", so, re.S)
    assert m, "rgFont"
    glyphs = []
    for row in re.findall(r"\{([^{}]*)\}", m.group(1)):
        codes = []
        for tok in re.findall(r"P\(\s*\d+\s*,\s*\d+\s*\)|FONT_UP|FONT_LAST", row):
            if tok == "FONT_UP":
                codes.append(0xFE)
            elif tok == "FONT_LAST":
                codes.append(0xFF)
            else:
                x, y = [int(v) for v in re.findall(r"\d+", tok)]
                codes.append(((x & 15) << 4) | (y & 15))
        assert 1 <= len(codes) <= 8, row
        glyphs.append(codes + [0] * (8 - len(codes)))
    assert len(glyphs) == 95 + 11, len(glyphs)
    tables["rgFont"] = np.array(glyphs, dtype=np.int32).reshape(-1)

    # ---- write tables.bin: magic, n, directory {name[40], dtype u32, count u32, offset u64}, payload (8B aligned)
    names = sorted(tables)
    DT = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int32): 2}
    dir_size = 16 + len(names) * 56
    payload = b""
    entries = []
    for nme in names:
        a = np.ascontiguousarray(tables[nme])
        while (dir_size + len(payload)) % 8:
            payload += b"\0"
        entries.append((nme, DT[a.dtype], a.size, dir_size + len(payload)))
        payload += a.tobytes()
    out = struct.pack("<8sII", b"RTTABLE1", len(names), 0)
    for nme, dt, cnt, off in entries:
        out += struct.pack("<40sIIQ", nme.encode(), dt, cnt, off)
    assert len(out) == dir_size
    out += payload
    p = os.path.join(ROOT, "ray-tracer_amd", "data", "tables.bin")
    with open(p, "wb") as f:
        f.write(out)
    print("wrote", p, len(out), "bytes,", len(names), "tables")
    g = os.path.join(ROOT, "tests", "golden", "quasi_index_tables.json")
    with open(g, "w") as f:
        json.dump({"source": "checked-in outputs of BuildQuasiIndex: LowDiscrepancy.cpp:262-333, BlueNoiseTile.cpp:1607-2417",
                   "tables": golden}, f, separators=(",", ":"))
    print("wrote", g)

if __name__ == "__main__":
    main()
