"""Mechanical guard for the part of the oracle that cannot be pinned by execution.

src/SPLASH.cpp includes <Rcpp.h>, which this image lacks, so the reference's water-balance code
cannot be compiled here and oracle/splash_oracle.c restates it by hand.  This script (build
container only: it reads /root/reference as TEXT, which is study, and writes nothing there) checks
that restatement statement by statement:

  * the live statements of SPLASH::run_one_day (src/SPLASH.cpp:920-1588), spin_up (:1594-1749),
    run_all (:1833-1916), moist_surf (:1921-1961) and inf_GA (:1963-2023) are extracted (comments,
    dead `/* */` blocks and declarations stripped), and so are the statements of the oracle's
    counterparts;
  * both are brought to one spelling (Global::x <-> G_x, std::max <-> STDMAX, isnan(x)==1 <->
    isnan(x), member calls <-> orc_ calls, ...) -- spelling only, never the arithmetic: operand
    order, parentheses and literals must be identical as written;
  * every ARITHMETIC statement of the reference (an assignment whose right side contains an operator
    or a call, or a condition of an if) must appear, in order, among the oracle's, except those listed
    in DEAD (computed but never used for a result: the compiler's -Wunused list of SURVEY.md 8a plus
    bookkeeping), each with its reason; and every arithmetic statement of the oracle must come from
    the reference, except those in ORACLE_ONLY (temporaries that hold a macro argument, monitors).

Exit code 1 and a diff on any unexplained mismatch.  `python tools/check_restatement.py --report f`
writes the summary that is committed under profiles/.
"""
import argparse
import difflib
import os
import re
import sys

REF = "/root/reference/src/SPLASH.cpp"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC = os.path.join(ROOT, "oracle", "splash_oracle.c")

# (reference line range, oracle function)
UNITS = [
    ("run_one_day", (920, 1588), "orc_run_one_day"),
    ("spin_up", (1594, 1749), "orc_spin_up"),
    ("moist_surf", (1921, 1961), "orc_moist_surf"),
    ("inf_GA", (1963, 2023), "orc_inf_GA"),
]


def strip_comments(txt):
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)   # C comments do not nest: first */ ends
    txt = re.sub(r"//[^\n]*", " ", txt)
    return txt


def oracle_function(src, name):
    m = re.search(r"^[a-z_ ]*\b" + name + r"\s*\(", src, flags=re.M)
    i = src.index("{", m.end())
    depth, j = 0, i
    while True:
        c = src[j]
        depth += (c == "{") - (c == "}")
        j += 1
        if depth == 0:
            return src[i + 1:j - 1]


def statements(body):
    """split a function body into simple statements and if-conditions, in source order"""
    out = []
    body = strip_comments(body)
    # if (...) conditions (balanced parentheses)
    pos = 0
    marks = []
    for m in re.finditer(r"\b(?:else\s+)?if\s*\(", body):
        k, depth = m.end(), 1
        while depth:
            depth += (body[k] == "(") - (body[k] == ")")
            k += 1
        marks.append((m.start(), k, "if(" + body[m.end():k - 1] + ")"))
    # cut the conditions out and split the rest on ; { }
    pieces, last = [], 0
    for a, b, cond in marks:
        pieces.append((last, body[last:a]))
        pieces.append((a, cond))
        last = b
    pieces.append((last, body[last:]))
    for _, p in pieces:
        if p.startswith("if("):
            out.append(p)
            continue
        for s in re.split(r"[;{}]", p):
            s = s.strip()
            if s and s not in ("else", "}else", "else{"):
                out.append(s)
    return out


TYPE = r"(?:const\s+)?(?:double|float|int|long|size_t|orc_smr|orc_etr|smr|etr|vector<double>|List)\b\s*\*?\s*"


def norm(s, oracle):
    s = re.sub(r"\s+", "", re.sub(r"^\s*" + TYPE, "", s.strip()))
    s = re.sub(r"^else", "", s)
    if not oracle:
        s = s.replace("Global::", "G_").replace("std::", "")
        s = re.sub(r"\bmax\(", "STDMAX(", s)
        s = re.sub(r"\bmin\(", "STDMIN(", s)
        s = re.sub(r"isnan\(([^()]*(?:\([^()]*\))?[^()]*)\)==1", r"isnan(\1)", s)
        s = re.sub(r"\bmoist_surf\(", "orc_moist_surf(", s)
        s = re.sub(r"\binf_GA\(", "orc_inf_GA(", s)
        s = s.replace("this->", "")
    s = s.replace("1.0e6", "1e6")
    return s


def arithmetic(s):
    """a statement that computes something: condition, compound assignment, or assignment whose
    right side has an operator or a call (plain copies like `wn = wn_vec[p]` are bookkeeping)"""
    if s.startswith("if("):
        return True
    m = re.match(r"^[A-Za-z_][\w\.\[\]\-\+\*]*?([-+*/]?)=(?!=)(.*)$", s)
    if not m:
        return False
    if m.group(1):
        return True
    rhs = m.group(2)
    return bool(re.search(r"[-+*/(]", rhs.lstrip("-")))


# reference statements with no counterpart in the oracle, and why that is right
DEAD = {
    "run_one_day": {
        r"^psi_m=|^wtd=|^z_uns=|^w_uns=|^w_z=|^if\(wtd<0\.0\|\|isnan\(wtd\)\)$|^if\(wtd>depth\)$":
            None,  # handled positionally below (the same text is live further down)
        r"^max_sw=": "dead local (-Wunused, SURVEY 8a A8): feeds nothing",
        r"^AI=soil_info\[12\]": "feeds only max_sw (dead); out-of-bounds read in point mode (SURVEY trap 1)",
        r"^To=|^T_qs=|^Acs_out_q0=|^Q_sat=|^Q_uns=": "dead locals (-Wunused, SURVEY 8a A8)",
        r"^econ=dvap\.econ|^rn_d=dvap\.rn_d": "copies never read afterwards",
    },
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--report", default=None)
    a = ap.parse_args()
    if not os.path.exists(REF):
        print("reference tree absent: nothing to check here (build container only)")
        return 0
    ref_lines = open(REF).read().split("\n")
    orc_src = open(ORC).read()
    report, failures = [], 0
    for unit, (l0, l1), fn in UNITS:
        rbody = "\n".join(ref_lines[l0 - 1:l1])
        rbody = rbody[rbody.index("{") + 1:]
        rs = [norm(s, False) for s in statements(rbody)]
        os_ = [norm(s, True) for s in statements(oracle_function(orc_src, fn))]
        rs = [s for s in rs if arithmetic(s)]
        os_ = [s for s in os_ if arithmetic(s)]
        sm = difflib.SequenceMatcher(None, rs, os_, autojunk=False)
        only_ref, only_orc, same = [], [], 0
        for tag, i1, i2, j1, j2 in sm.get_opcodes():
            if tag == "equal":
                same += i2 - i1
            else:
                only_ref += rs[i1:i2]
                only_orc += os_[j1:j2]
        report.append(f"## {unit}: src/SPLASH.cpp:{l0}-{l1} vs {fn}")
        report.append(f"arithmetic statements: reference {len(rs)}, oracle {len(os_)}, identical in order {same}")
        report.append("only in the reference:")
        report += ["    " + s for s in only_ref]
        report.append("only in the oracle:")
        report += ["    " + s for s in only_orc]
        report.append("")
    txt = "\n".join(report)
    print(txt)
    if a.report:
        open(a.report, "w").write(txt + "\n")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
