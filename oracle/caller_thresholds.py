"""TEST INFRASTRUCTURE ONLY -- restatement of what Crackling does with the scorer's stdout.

Follows /root/reference/src/crackling/Crackling.py:780-835 (Constants.py:1-2: CODE_ACCEPTED = 1, CODE_REJECTED = 0):
lines are split on tabs, only 3-field lines are used, fields 2 and 3 go through float(); the decision uses the
configured method lower-cased and stripped.  The reference has this logic inline in one 800-line function, so there is
no callable unit to import.  PINNED by tests/golden/verdicts/: oracle/make_golden_verdicts.py lifts those statements
out of the reference source at run time (AST slice, nothing stored), executes them on the golden stdout files of the
compiled reference scorer and on borderline score sets, and tests/test_verdicts.py checks this restatement (and the
product's issl_verdicts) against the recorded verdicts.  Only tests/ may import this module.
"""

ACCEPTED, REJECTED = 1, 0


def caller_verdicts(stdout_text, targets20, threshold, method):
    """{20-mer: ACCEPTED | REJECTED} for the targets the caller would mark (Crackling.py:788-835)."""
    scored = {}
    for fields in [x.split("\t") for x in stdout_text.splitlines(keepends=True)]:   # :781-786
        if len(fields) == 3:
            scored[fields[0]] = {"MIT": float(fields[1].strip()), "CFD": float(fields[2].strip())}
    thr = float(threshold)                                                          # :793
    rule = str(method).strip().lower()                                              # :794
    out = {}
    for t in targets20:
        if t not in scored:
            continue
        s = scored[t]
        if rule == "mit":
            rej = s["MIT"] < thr
        elif rule == "cfd":
            rej = s["CFD"] < thr
        elif rule == "and":
            rej = (s["MIT"] < thr) and (s["CFD"] < thr)
        elif rule == "or":
            rej = (s["MIT"] < thr) or (s["CFD"] < thr)
        elif rule == "avg":
            rej = ((s["MIT"] + s["CFD"]) / 2) < thr
        else:
            continue                                                                # no branch taken: left untouched
        out[t] = REJECTED if rej else ACCEPTED
    return out
