"""Minimal reader for gzip'd R `save()` files (RDX2 / XDR serialisation, format version 2).

Used once, in the build container, to turn the reference's bundled example data
(data/*.RData) into plain CSV/JSON/NPZ fixtures under tests/golden/ -- there is no R
interpreter in the image.  Only the node types that occur in those files are handled.
"""
import gzip
import struct

import numpy as np

NILVALUE_SXP, GLOBALENV_SXP, REFSXP = 254, 253, 255
EMPTYENV_SXP, BASEENV_SXP, MISSINGARG_SXP, BASENAMESPACE_SXP = 242, 241, 251, 247
NAMESPACESXP, PACKAGESXP, PERSISTSXP = 249, 250, 248
ATTRLISTSXP, ATTRLANGSXP, BCREPDEF, BCREPREF = 239, 240, 244, 243
SYMSXP, LISTSXP, CLOSXP, ENVSXP, PROMSXP, LANGSXP = 1, 2, 3, 4, 5, 6
SPECIALSXP, BUILTINSXP, CHARSXP, LGLSXP, INTSXP, REALSXP = 7, 8, 9, 10, 13, 14
CPLXSXP, STRSXP, DOTSXP, VECSXP, EXPRSXP, BCODESXP, EXTPTRSXP, RAWSXP, S4SXP = \
    15, 16, 17, 19, 20, 21, 22, 24, 25


class RObj:
    """value + attributes (attributes is a dict; pairlists become list of (tag, value))."""

    def __init__(self, value, attrs=None, kind=None):
        self.value, self.attrs, self.kind = value, attrs or {}, kind

    def __repr__(self):
        return f"RObj({self.kind}, attrs={list(self.attrs)})"


class Reader:
    def __init__(self, buf):
        self.b, self.p, self.refs = buf, 0, []

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.p)[0]
        self.p += 4
        return v

    def take(self, n):
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def length(self):
        n = self.i32()
        if n == -1:
            hi, lo = self.i32(), self.i32()
            n = (hi << 32) + lo
        return n

    def pairlist_to_attrs(self, pl):
        return {t: v for t, v in pl} if pl else {}

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
        if t == NILVALUE_SXP:
            return None
        if t in (GLOBALENV_SXP, EMPTYENV_SXP, BASEENV_SXP, MISSINGARG_SXP, BASENAMESPACE_SXP):
            return RObj(None, kind=f"env{t}")
        if t == REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t in (NAMESPACESXP, PACKAGESXP, PERSISTSXP):
            self.i32()
            n = self.i32()
            names = [self.item() for _ in range(n)]
            o = RObj(names, kind="namespace")
            self.refs.append(o)
            return o
        if t == SYMSXP:
            name = self.item()
            self.refs.append(name)
            return name
        if t in (LISTSXP, LANGSXP, CLOSXP, PROMSXP, DOTSXP, ATTRLISTSXP, ATTRLANGSXP):
            out = []
            while True:  # iterate the cdr chain without recursion
                attrs = self.item() if (has_attr or t in (ATTRLISTSXP, ATTRLANGSXP)) else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                flags = self.i32()
                t2 = flags & 0xFF
                if t2 == NILVALUE_SXP:
                    break
                if t2 not in (LISTSXP, LANGSXP, ATTRLISTSXP, ATTRLANGSXP):
                    self.p -= 4
                    out.append(("__cdr__", self.item()))
                    break
                has_attr, has_tag, t = bool(flags & 0x200), bool(flags & 0x400), t2
                _ = attrs
            return out
        if t == ENVSXP:
            o = RObj(None, kind="env")
            self.refs.append(o)
            self.i32()
            enclos, frame, hashtab, attrs = self.item(), self.item(), self.item(), self.item()
            o.value = (frame, hashtab)
            return o
        if t == EXTPTRSXP:
            o = RObj(None, kind="extptr")
            self.refs.append(o)
            o.value = (self.item(), self.item())
            if has_attr:
                o.attrs = self.pairlist_to_attrs(self.item())
            return o
        if t in (SPECIALSXP, BUILTINSXP):
            n = self.i32()
            return RObj(self.take(n).decode(), kind="builtin")
        if t == CHARSXP:
            n = self.i32()
            return None if n == -1 else self.take(n).decode("latin-1")
        if t == LGLSXP or t == INTSXP:
            n = self.length()
            v = np.frombuffer(self.take(4 * n), dtype=">i4").astype(np.int32)
            o = RObj(v, kind="lgl" if t == LGLSXP else "int")
        elif t == REALSXP:
            n = self.length()
            o = RObj(np.frombuffer(self.take(8 * n), dtype=">f8").astype(np.float64), kind="real")
        elif t == CPLXSXP:
            n = self.length()
            o = RObj(np.frombuffer(self.take(16 * n), dtype=">c16"), kind="cplx")
        elif t == STRSXP:
            n = self.length()
            o = RObj([self.item() for _ in range(n)], kind="str")
        elif t in (VECSXP, EXPRSXP):
            n = self.length()
            o = RObj([self.item() for _ in range(n)], kind="list")
        elif t == RAWSXP:
            n = self.length()
            o = RObj(self.take(n), kind="raw")
        elif t == S4SXP:
            o = RObj(None, kind="S4")
        elif t == BCODESXP:  # compiled closures inside S4 objects (SA_cru.RData): parsed to skip over
            reps = [None] * self.i32()
            o = RObj(self.bc1(reps), kind="bytecode")
        else:
            raise NotImplementedError(f"SEXP type {t} at {self.p}")
        if has_attr:
            o.attrs = self.pairlist_to_attrs(self.item())
        return o


def _bc1(self, reps):
    code = self.item()
    n = self.i32()
    consts = []
    for _ in range(n):
        t = self.i32()
        if t == BCODESXP:
            consts.append(self.bc1(reps))
        elif t in (LANGSXP, LISTSXP, BCREPDEF, BCREPREF, ATTRLANGSXP, ATTRLISTSXP):
            consts.append(self.bclang(t, reps))
        else:
            consts.append(self.item())
    return (code, consts)


def _bclang(self, t, reps):
    if t == BCREPREF:
        return reps[self.i32()]
    if t in (BCREPDEF, LANGSXP, LISTSXP, ATTRLANGSXP, ATTRLISTSXP):
        pos = -1
        if t == BCREPDEF:
            pos, t = self.i32(), self.i32()
        node = RObj(None, kind="bclang")
        if pos >= 0:
            reps[pos] = node
        attr = self.item() if t in (ATTRLANGSXP, ATTRLISTSXP) else None
        tag = self.item()
        car = self.bclang(self.i32(), reps)
        cdr = self.bclang(self.i32(), reps)
        node.value = (tag, car, cdr, attr)
        return node
    return self.item()  # the type word was a pad


Reader.bc1, Reader.bclang = _bc1, _bclang


def load(path):
    raw = gzip.open(path, "rb").read()
    assert raw[:5] == b"RDX2\n", raw[:5]
    assert raw[5:7] == b"X\n"
    r = Reader(raw)
    r.p = 7
    r.i32(), r.i32(), r.i32()  # format version, writer version, min reader version
    top = r.item()  # pairlist of (name, value)
    return {k: v for k, v in top}


def named_list(o):
    names = o.attrs.get("names")
    return dict(zip(names.value, o.value)) if names is not None else o.value
