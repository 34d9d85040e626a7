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
    # spin_up's day step (called at :1658, :1674, :1715, :1729): a second copy of run_one_day with other
    # names for the EVAP result and the output struct, and without the bookkeeping of the public method
    ("quick_run", (150, 918), "orc_run_one_day"),
    ("spin_up", (1594, 1749), "orc_spin_up"),
    ("run_all", (1833, 1916), "run_all_strided"),
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


# Per-unit renames applied to the REFERENCE side before the comparison -- names only, plus the one
# literal quick_run spells differently from run_one_day (`Ksat_visc*24.0` at :505 vs `Ksat_visc*24` at
# :1281: the int literal is converted to the same double 24.0 by the usual arithmetic conversions).
RENAMES = {
    "quick_run": [(r"\bdn\.", "dvap."), (r"^dn=", "dvap="), (r"\bdsm\b", "dsoil"),
                  (r"\(infi/\(Ksat_visc\*24\.0\)\)", "(infi/(Ksat_visc*24))")],
}


def norm(s, oracle, unit=None):
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
        for pat, rep in RENAMES.get(unit, ()):
            s = re.sub(pat, rep, s)
    s = s.replace("1.0e6", "1e6").replace("*fstride", "")
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
    rhs = re.sub(r"\[[^\]]*\]", "[]", m.group(2))
    return bool(re.search(r"[-+*/(]", rhs.lstrip("-")))


# Statements on one side only, and why that is right.  Anything not listed fails the check.
DEAD = [  # reference only
    (r"^(psi_m|wtd|z_uns|sat_swc|w_uns_z|w_uns_0|w_uns|w_z)=|^if\(wtd<0\.0\|\|isnan\(wtd\)\)$|^if\(wtd>depth\)$|^if\(wtd<=2\.0\)$",
     "water held above 2 m, src/SPLASH.cpp:999-1022: computed, never used for a result (SURVEY 8a A8)"),
    (r"^max_sw=|^AI=", "dead local (-Wunused); AI = soil_info[12] only feeds it (SURVEY trap 1)"),
    (r"^dvap=evap\.get_vals\(\)", "member copy of the EVAP result (the oracle passes a struct)"),
    (r"^(Acs_out_q0|psi_qs|wtd_qs|T_qs|To|To_sat|Q_uns|Q_sat)=|^To_uns\*=|^if\(wtd_qs", "dead locals "
     "(-Wunused list of SURVEY 8a A8) and what only feeds them"),
    (r"^Q=\(T\*Ai\)/1000$", "second evaluation after the last transmittance: never read (:1546)"),
    (r"^theta_m=STDMAX\(Wmax/\(depth\*1000\.0\),theta_i\)$|^ro_h=STDMAX\(inflow-infi,0\.0\)$|"
     r"^tdrain_out=STDMAX\(\(td\+t_drain\)/2,0\.0\)$|^qin_nday=STDMAX\(STDMAX\(q_in_o,q_in_f\),0\.0\)$",
     "std::max of an expression: the oracle's macro evaluates its arguments twice, so it names them first"),
    (r"^sm-=\(Drain_out\)$", "Drain_out = T (:1547), a plain copy"),
    (r"^(theta_s|theta_r)=|^(start_wtd|end_wtd|diff_swe|diff_wtd)=|^if\(diff_swe<0\)|^if\(diff_wtd<0\)",
     "spin_up: water-table and snow differences are computed but never gate the loop (:1697)"),
    (r"^if\(i==0\)$|^(wn|snow|qin|td|nd)=\w+\[", "spin_up: previous-day bookkeeping (wrap-around read at i == 0)"),
    (r"^(wn|swe|qin|td|nd)=\w|^\w+\[i\]=|^if\(i==0\)$|^n_end=", "run_all: series bookkeeping (first day from the *_last arguments, then the previous element); netr = rn_d/1e6 (:1907) is o[7] in the oracle"),
]
ORACLE_ONLY = [
    (r"^wm_th=Wmax/\(depth\*1000\.0\)$|^theta_m=STDMAX\(wm_th,theta_i\)$|^ro_h_arg=inflow-infi$|"
     r"^ro_h=STDMAX\(ro_h_arg,0\.0\)$|^td_arg=\(td\+t_drain\)/2$|^tdrain_out=STDMAX\(td_arg,0\.0\)$|"
     r"^mx=STDMAX\(q_in_o,q_in_f\)$|^qin_nday=STDMAX\(mx,0\.0\)$", "named macro arguments (see above)"),
    (r"^sm-=\(T\)$", "Drain_out = T"),
    (r"^if\(Au>0\.0&&wtd>=depth&&", "conditioning monitors (orc_fragile_events / orc_critical_events): no effect on results"),
    (r"^if\(dvap_out\)$|^if\((sm|snow|qin|tdrain|ro|snwage|pet)_o\)$", "optional outputs"),
    (r"^\w+_vec=|^nb=sizeof|^p=\(i==0\)", "spin_up: one malloc'd block instead of std::vectors; previous-day index"),
    (r"^if\(g_spin_trace&&|^t5=g_spin_trace\+|^t5\[\d\]=", "spin_up: analysis hook recording the state after day 1 of a pass (tools/tail_periods.py): no effect on results"),
    (r"^k=\(size_t\)i\*ostride$|^if\(o\[\d+\]\)$|^if\(state_end\)$|^(wn|swe|qin|td|nd)=\w|^o\[7\]\[k\]=dv\.rn_d/1e6$", "run_all: strided outputs, optional series, the carried state; o[7] = netr_vec[i] = dvap.rn_d/1e6 (:1907)"),
]


def day_calls(stmts, oracle):
    """the daily-step calls of a driver loop, arguments only, outputs and (lat, elv) removed:
    quick_run(n, y, wn, sw_in, tc, pn, dsm, slop, asp, snow, snowfall, soil_info, qin, td, nd) <->
    orc_run_one_day(lat, elv, n, y, wn, sw_in, tc, pn, slop, asp, snow, snowfall, soil_info, qin, td, nd, &dsm, NULL)"""
    out = []
    for s in stmts:
        m = re.match(r"^(?:orc_run_one_day|quick_run|run_one_day)\((.*)\)$", s)
        if not m:
            continue
        args, depth, cur = [], 0, ""
        for ch in m.group(1):
            if ch == "," and depth == 0:
                args.append(cur)
                cur = ""
            else:
                depth += (ch in "([") - (ch in ")]")
                cur += ch
        args.append(cur)
        if oracle:
            args = args[2:-2]
        else:
            args = [a for a in args if a not in ("dsm", "dsoil")]
        out.append(",".join(x.replace("(i+1)", "i+1").replace("(i-1)", "i-1").replace("swe", "snow")
                            for x in args))
    return out


def explained(stmt, table):
    for pat, why in table:
        if re.search(pat, stmt):
            return why
    return None


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
        rs = [norm(s, False, unit) for s in statements(rbody)]
        os_ = [norm(s, True) for s in statements(oracle_function(orc_src, fn))]
        rc, oc = day_calls(rs, False), day_calls(os_, True)
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
        for title, items, table in (("only in the reference", only_ref, DEAD), ("only in the oracle", only_orc, ORACLE_ONLY)):
            report.append(title + ":")
            for st in items:
                why = explained(st, table)
                if why is None:
                    failures += 1
                report.append(f"    {st}\n        -> {why or 'UNEXPLAINED'}")
        if rc or oc:
            # the oracle reads the previous day's state into wn, snow, ... through the index p
            same_calls = [x == y for x, y in zip(rc, oc)]
            report.append(f"daily-step calls: reference {len(rc)}, oracle {len(oc)}, identical argument lists "
                          f"{sum(same_calls)}")
            for x, y in zip(rc, oc):
                if x != y:
                    failures += 1
                    report.append(f"    reference ({x})\n    oracle    ({y})")
            failures += abs(len(rc) - len(oc))
        report.append("")
    report.append(f"unexplained differences: {failures}")
    txt = "\n".join(report)
    print(txt)
    if a.report:
        open(a.report, "w").write(txt + "\n")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
