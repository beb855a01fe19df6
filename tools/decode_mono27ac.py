#!/usr/bin/env python3
"""Decode the reference's bundled data set data/Mono27ac.RData (XZ-compressed R
serialization, format RDX2/XDR) into the bedGraph text fixture used by the tests.

A fixture is data: this script only reads the reference's *data file* and writes
tests/golden/Mono27ac.bedGraph (chrom, chromStart, chromEnd, count; tab separated) and
tests/golden/Mono27ac.labels.bed.  SURVEY.md section 8c records the expected sha256 of the
coverage dump: 67f3ff8b786b164548bb9d5961d551cfa1ea865a42a3cd3a5da747103977ac97.

usage: python tools/decode_mono27ac.py /root/reference/data/Mono27ac.RData tests/golden
"""
import hashlib
import lzma
import struct
import sys


class Reader:
    def __init__(self, buf):
        self.b = buf
        self.i = 0
        self.refs = []

    def int(self):
        v = struct.unpack_from(">i", self.b, self.i)[0]
        self.i += 4
        return v

    def dbl(self):
        v = struct.unpack_from(">d", self.b, self.i)[0]
        self.i += 8
        return v

    def bytes(self, n):
        v = self.b[self.i:self.i + n]
        self.i += n
        return v

    def item(self):
        flags = self.int()
        t = flags & 0xFF
        has_attr = bool(flags & 0x200)
        has_tag = bool(flags & 0x400)
        if t == 254:      # NILVALUE_SXP
            return None
        if t == 255:      # REFSXP
            return self.refs[(flags >> 8) - 1]
        if t == 253:      # R_EmptyEnv
            return "<emptyenv>"
        if t == 242:      # R_GlobalEnv
            return "<globalenv>"
        if t == 1:        # SYMSXP
            name = self.item()
            self.refs.append(name)
            return name
        if t == 2:        # LISTSXP (pairlist)
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                nxt = self.int()
                nt = nxt & 0xFF
                if nt == 254:
                    return out
                if nt != 2:
                    raise ValueError("unexpected pairlist tail type %d" % nt)
                has_attr = bool(nxt & 0x200)
                has_tag = bool(nxt & 0x400)
        if t == 9:        # CHARSXP
            n = self.int()
            return None if n == -1 else self.bytes(n).decode("utf-8")
        if t == 10 or t == 13:   # LGLSXP / INTSXP
            n = self.int()
            v = list(struct.unpack_from(">%di" % n, self.b, self.i))
            self.i += 4 * n
            val = v
        elif t == 14:     # REALSXP
            n = self.int()
            v = list(struct.unpack_from(">%dd" % n, self.b, self.i))
            self.i += 8 * n
            val = v
        elif t == 16:     # STRSXP
            n = self.int()
            val = [self.item() for _ in range(n)]
        elif t == 19:     # VECSXP
            n = self.int()
            val = [self.item() for _ in range(n)]
        elif t == 22:     # EXTPTRSXP (data.table's .internal.selfref)
            self.refs.append("<extptr>")
            self.item()
            self.item()
            val = "<extptr>"
        else:
            raise ValueError("unsupported SEXP type %d at %d" % (t, self.i))
        attrs = dict(self.item()) if has_attr else {}
        return {"v": val, "a": attrs} if attrs else val


def main():
    src, outdir = sys.argv[1], sys.argv[2]
    raw = lzma.decompress(open(src, "rb").read())
    assert raw[:5] == b"RDX2\n", raw[:5]
    r = Reader(raw[5:])
    assert r.bytes(2) == b"X\n"
    r.int(); r.int(); r.int()          # format version, writer version, min reader version
    top = dict(r.item())               # pairlist: name -> object
    mono = top["Mono27ac"]
    names = mono["a"]["names"]
    parts = dict(zip(names if isinstance(names, list) else names["v"], mono["v"]))

    def table(obj):
        cols = obj["a"]["names"]
        cols = cols if isinstance(cols, list) else cols["v"]
        out = {}
        for c, v in zip(cols, obj["v"]):
            if isinstance(v, dict):
                if "levels" in v["a"]:      # factor
                    lev = v["a"]["levels"]
                    lev = lev if isinstance(lev, list) else lev["v"]
                    v = [lev[i - 1] for i in v["v"]]
                else:
                    v = v["v"]
            out[c] = v
        return cols, out

    cols, cov = table(parts["coverage"])
    assert cols == ["chrom", "chromStart", "chromEnd", "count"], cols
    text = "".join("%s\t%d\t%d\t%d\n" % row for row in
                   zip(cov["chrom"], cov["chromStart"], cov["chromEnd"], cov["count"]))
    open(outdir + "/Mono27ac.bedGraph", "w").write(text)
    lcols, lab = table(parts["labels"])
    ltext = "".join("\t".join(str(lab[c][i]) for c in lcols) + "\n"
                    for i in range(len(lab[lcols[0]])))
    open(outdir + "/Mono27ac.labels.bed", "w").write(ltext)
    print("coverage rows", len(cov["count"]), "sha256", hashlib.sha256(text.encode()).hexdigest())
    print("labels rows", len(lab[lcols[0]]), lcols)


if __name__ == "__main__":
    main()
