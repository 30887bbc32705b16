"""Token statistics of a MinLZ block (format per SPEC.md:68-266 as restated in oracle/minlz_oracle.c decode)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def walk(enc):
    b = memoryview(enc)
    assert b[0] == 0
    s = 1
    n = 0; sh = 0
    while True:
        c = b[s]; s += 1
        n |= (c & 0x7f) << sh; sh += 7
        if c < 0x80:
            break
    st = dict(lit_bytes=0, lit_runs=0, rep=0, rep_bytes=0, c1=0, c1_bytes=0, c2=0, c2_bytes=0, f2=0, f2_bytes=0, c3=0, c3_bytes=0,
              hdr_bytes=0)
    offs = []; lens = []
    E = len(b)
    while s < E:
        t = b[s]; tag = t & 3
        if tag == 0:
            x = t >> 3; h = 1
            if x < 29: ln = x + 1
            else:
                k = x - 28
                ln = 30 + int.from_bytes(b[s + 1:s + 1 + k], "little"); h += k
            if t & 4:
                st["rep"] += 1; st["rep_bytes"] += ln; st["hdr_bytes"] += h; s += h
                lens.append(ln); offs.append(0)
            else:
                st["lit_runs"] += 1; st["lit_bytes"] += ln; st["hdr_bytes"] += h; s += h + ln
        elif tag == 1:
            v = b[s] | (b[s + 1] << 8); ln = (v >> 2) & 15; h = 2
            if ln == 15: ln = b[s + 2] + 18; h = 3
            else: ln += 4
            st["c1"] += 1; st["c1_bytes"] += ln; st["hdr_bytes"] += h; s += h
            lens.append(ln); offs.append((v >> 6) + 1)
        elif tag == 2:
            ln = t >> 2; h = 3
            off = (b[s + 1] | (b[s + 2] << 8)) + 64
            if ln <= 60: ln += 4
            else:
                k = ln - 60
                ln = 64 + int.from_bytes(b[s + 3:s + 3 + k], "little"); h += k
            st["c2"] += 1; st["c2_bytes"] += ln; st["hdr_bytes"] += h; s += h
            lens.append(ln); offs.append(off)
        else:
            v = int.from_bytes(b[s:s + 4], "little")
            if not (v & 4):
                lits = ((v >> 3) & 3) + 1; ln = 4 + ((v >> 5) & 7); off = ((v >> 8) & 0xffff) + 64
                st["f2"] += 1; st["f2_bytes"] += ln; st["lit_bytes"] += lits; st["hdr_bytes"] += 3; s += 3 + lits
            else:
                lits = (v >> 3) & 3; lc = (v >> 5) & 63; off = (v >> 11) + 65536; h = 4
                if lc <= 60: ln = lc + 4
                else:
                    k = lc - 60
                    ln = 64 + int.from_bytes(b[s + 4:s + 4 + k], "little"); h += k
                st["c3"] += 1; st["c3_bytes"] += ln; st["lit_bytes"] += lits; st["hdr_bytes"] += h; s += h + lits
            lens.append(ln); offs.append(off)
    st["n"] = n; st["enc"] = len(b)
    return st, np.array(offs), np.array(lens)


def report(name, enc):
    st, offs, lens = walk(enc)
    n = st["n"]
    m = len(lens)
    print("%-12s enc %8d (%.4f)  lit %.4f hdr %.4f | matches %7d avg len %.2f | rep %d/%d c1 %d/%d c2 %d/%d f2 %d/%d c3 %d/%d" % (
        name, st["enc"], st["enc"] / n, st["lit_bytes"] / n, st["hdr_bytes"] / n, m, lens.mean(),
        st["rep"], st["rep_bytes"], st["c1"], st["c1_bytes"], st["c2"], st["c2_bytes"], st["f2"], st["f2_bytes"], st["c3"], st["c3_bytes"]))
    real = offs > 0
    for lo, hi in ((1, 1024), (1024, 32768), (32768, 65600), (65600, 1 << 20), (1 << 20, 1 << 24)):
        k = real & (offs >= lo) & (offs < hi)
        print("     off [%7d,%8d): %7d matches, %8d bytes (%.3f of N), avg len %.1f" % (lo, hi, k.sum(), lens[k].sum(), lens[k].sum() / n, lens[k].mean() if k.any() else 0))
    return st


if __name__ == "__main__":
    import oracle as O
    from minlz_amd import synth
    kind = sys.argv[1] if len(sys.argv) > 1 else "text"
    d = synth.text_like(8 << 20, 1) if kind == "text" else synth.json_like(8 << 20)
    for lv in (1, 2):
        report("oracle L%d" % lv, O.encode(d, lv))
        p = "gpurun_out/enc_%s_l%d.bin" % (kind, lv)
        if os.path.exists(p):
            report("gpu L%d" % lv, open(p, "rb").read())
