#!/usr/bin/env python3
"""Golden verdicts from the reference caller's own loop (tests/golden/verdicts/).

Run in the build container only (needs /root/reference):

    python oracle/make_golden_verdicts.py

What Crackling does with the scorer's stdout -- parse the 3-field lines, compare with the threshold under the
configured method -- is inline in one long function (src/crackling/Crackling.py:780-835), so there is nothing to
import and call.  This script lifts exactly those statements out of the reference source AT RUN TIME: it parses the
file with `ast`, finds the `with open(configMngr['offtargetscore']['output'], 'r')` block that reads the scorer's
output, takes it together with the statement before it (`targetsScored = {}`) and the two after it (`failedCount = 0`
and the `for target23 in pageCandidateGuides` loop), compiles those four nodes and executes them with a stub config
and a stub candidate table.  CODE_ACCEPTED / CODE_REJECTED are read from the literal assignments in Constants.py.
Nothing of the reference's text is stored: only inputs and the verdicts it produced.

Cases:
  * every stdout of the compiled reference scorer in tests/golden/{uniform,clustered,edge}/expected.json
    (stored as one character per guide in guide order: '1' accepted, '0' rejected, '-' left untouched), also under
    method strings the scorer does not know but the caller lower-cases ("AND", " avg", "Mit");
  * synthetic score pairs around the thresholds whose 6-decimal text lands on either side of them (the doubles are
    stored as hex so that the test can hand the very same values to issl_verdicts)."""
import ast, json, pathlib, random, sys, tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
REF = pathlib.Path("/root/reference/src/crackling")
GOLD = ROOT / "tests" / "golden"


def lift_caller_loop():
    tree = ast.parse((REF / "Crackling.py").read_text())

    def reads_scorer_output(node):
        if not isinstance(node, ast.With) or len(node.items) != 1:
            return False
        call = node.items[0].context_expr
        return (isinstance(call, ast.Call) and getattr(call.func, "id", None) == "open" and len(call.args) == 2
                and ast.unparse(call.args[0]) == "configMngr['offtargetscore']['output']"
                and getattr(call.args[1], "value", None) == "r")

    for parent in ast.walk(tree):
        body = getattr(parent, "body", None)
        if not isinstance(body, list):
            continue
        for i, node in enumerate(body):
            if reads_scorer_output(node):
                nodes = body[i - 1:i + 3]
                kinds = [type(n).__name__ for n in nodes]
                assert kinds == ["Assign", "With", "Assign", "For"], kinds
                assert ast.unparse(nodes[0].targets[0]) == "targetsScored" and ast.unparse(nodes[3].iter) == "pageCandidateGuides"
                lo, hi = nodes[0].lineno, nodes[3].end_lineno
                mod = ast.Module(body=nodes, type_ignores=[])
                return compile(mod, f"<Crackling.py:{lo}-{hi}>", "exec"), (lo, hi)
    raise SystemExit("the caller's loop was not found in the reference source")


def constants():
    out = {}
    for node in ast.parse((REF / "Constants.py").read_text()).body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") in ("CODE_ACCEPTED", "CODE_REJECTED"):
            out[node.targets[0].id] = ast.literal_eval(node.value)
    assert set(out) == {"CODE_ACCEPTED", "CODE_REJECTED"}
    return out


def run_caller(code, consts, stdout_text, targets20, threshold, method):
    """-> one character per target: '1' accepted, '0' rejected, '-' untouched."""
    with tempfile.NamedTemporaryFile("w", suffix=".output", delete=False) as f:
        f.write(stdout_text)
        path = f.name
    targets23 = [t + "AGG" for t in targets20]
    table = {t: {} for t in targets23}
    env = dict(consts)
    env.update({
        "configMngr": {"offtargetscore": {"output": path, "score-threshold": threshold, "method": method}},
        "pageCandidateGuides": targets23,
        "candidateGuides": table,
    })
    exec(code, env)
    pathlib.Path(path).unlink()
    chars = []
    for t in targets23:
        v = table[t].get("passedOffTargetScore")
        chars.append("-" if v is None else ("1" if v == consts["CODE_ACCEPTED"] else "0"))
    return "".join(chars)


def main():
    code, (lo, hi) = lift_caller_loop()
    consts = constants()
    out_dir = GOLD / "verdicts"
    out_dir.mkdir(parents=True, exist_ok=True)
    # 1. the reference scorer's own stdout
    cases = []
    for name in ("uniform", "clustered", "edge"):
        expected = json.loads((GOLD / name / "expected.json").read_text())
        for key, text in sorted(expected.items()):
            method, thr, _ = key.split("|")
            seqs = [line.split("\t")[0] for line in text.splitlines()]
            for cfg_method in {method, method.upper(), " " + method, method.capitalize()}:
                # the scorer was run with `method`; the caller is configured with the same string in Crackling, but its
                # lower-casing is part of the behaviour, so the other spellings are recorded against the same stdout
                cases.append({"golden": name, "key": key, "config_method": cfg_method, "threshold": thr,
                              "verdicts": run_caller(code, consts, text, seqs, thr, cfg_method)})
    (out_dir / "reference_stdout.json").write_text(json.dumps({"source_lines": [lo, hi], "cases": cases}, indent=0))
    # 2. borderline scores
    rnd = random.Random(20261004)
    letters = "ACGT"
    synth = []
    for thr in ("75", "0", "50", "99.9999995", "75.0000004"):
        t = float(thr)
        seqs, mits, cfds = [], [], []
        seen = set()
        while len(seqs) < 300:
            s = "".join(rnd.choice(letters) for _ in range(20))
            if s in seen:
                continue
            seen.add(s)
            seqs.append(s)
            edge = t + rnd.choice([-6e-7, -5e-7, -4e-7, -1e-7, 0.0, 1e-7, 4e-7, 5e-7, 6e-7, 1.1e-6, -1.1e-6])
            kind = rnd.randrange(4)
            mits.append(edge if kind in (0, 2) else rnd.uniform(0, 100))
            cfds.append(edge if kind in (1, 2) else rnd.uniform(0, 100))
        by_method = {}
        for method in ("and", "or", "avg", "mit", "cfd", "AND", " avg", "Mit", "xyz", ""):
            printed = method  # what the scorer is started with: it matches the method string exactly
            want_mit = printed in ("mit", "and", "or", "avg")
            want_cfd = printed in ("cfd", "and", "or", "avg")
            text = "".join(f"{s}\t{('%f' % m) if want_mit else '-1'}\t{('%f' % c) if want_cfd else '-1'}\n"
                           for s, m, c in zip(seqs, mits, cfds))
            by_method[method] = run_caller(code, consts, text, seqs, thr, method)
        synth.append({"threshold": thr, "seqs": seqs, "mit_hex": [m.hex() for m in mits], "cfd_hex": [c.hex() for c in cfds],
                      "verdicts_by_config_method": by_method})
    (out_dir / "borderline.json").write_text(json.dumps({"source_lines": [lo, hi], "cases": synth}, indent=0))
    allv = [c["verdicts"] for c in cases] + [v for c in synth for v in c["verdicts_by_config_method"].values()]
    n0, n1, nn = (sum(v.count(ch) for v in allv) for ch in "01-")
    print(f"lifted Crackling.py:{lo}-{hi}; {len(cases)} reference-stdout cases, {len(synth)} borderline score sets x 10 methods; "
          f"{n1} accepted, {n0} rejected, {nn} untouched")
    sys.exit(0 if n0 and n1 and nn else 1)


if __name__ == "__main__":
    main()
