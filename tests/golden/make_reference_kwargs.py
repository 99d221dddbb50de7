"""Extracts the keyword-argument NAMES of the reference functions on the hot path (data, not source) into
tests/golden/reference_kwargs.json.  Run in the build container, where /root/reference exists:
    python tests/golden/make_reference_kwargs.py"""
import json
import os
import re

REF = "/root/reference/src"
SITES = {  # function -> (file, 1-based line of its `function` statement)   
    "execute_range": ("NMFkExecute.jl", 178),
    "execute_single": ("NMFkExecute.jl", 236),
    "execute_run": ("NMFkExecute.jl", 483),
    "execute_singlerun_compute": ("NMFkExecute.jl", 729),
    "NMFmultiplicative": ("NMFkMultiplicative.jl", 24),
}


def kwnames(sig):
    sig = sig[sig.index(";") + 1:sig.rindex(")")]
    out, depth, cur = [], 0, ""
    for ch in sig:
        depth += ch in "({[" 
        depth -= ch in ")}]"
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    names = [re.match(r"\s*(\w+)", p).group(1) for p in out if p.strip() and not p.strip().endswith("...")]
    return names


if __name__ == "__main__":
    res = {}
    for fn, (file, line) in SITES.items():
        text = open(os.path.join(REF, file)).read().split("\n")[line - 1]
        assert text.lstrip().startswith("function"), (fn, text[:60])
        res[fn] = kwnames(text)
    fields = []
    lines = open(os.path.join(REF, "NMFkExecute.jl")).read().split("\n")
    i = next(k for k, l in enumerate(lines) if "struct ExecuteOptions" in l)
    for l in lines[i + 1:]:
        if l.strip() == "end":
            break
        fields.append(re.match(r"\s*(\w+)", l).group(1))
    res["ExecuteOptions"] = fields
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kwargs.json"), "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))
