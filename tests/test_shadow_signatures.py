"""Boundary conformance of the MATLAB overlay: a shadowing matlab/<name>.m must declare the output
list and the input argument ORDER of the reference function it shadows, because the reference's callers pass their
arguments by position (PP/imageMatching/imageMatchingPanoramaConComps.m:40,89 -> imageMatching(input, n, keypoints,
matchesAll, imagesProcessed); round 4 shipped that shadow with arguments 3 and 4 swapped).

  * tests/golden/reference_signatures.json is the table {name: {"outs": [...], "ins": [...], "ref": "file:line"}} of the
    reference's `function` lines and, for the three mex entry points, of the call sites that define their surface.  It
    travels with the repo; when /root/reference is present (the build container) the table is re-derived from the
    reference text and must equal the committed one.
  * every matlab/*.m with an entry in the table is compared with it, name by name and position by position;
  * every aps_mex('<cmd>', ...) call in matlab/*.m is checked against the argument count the gateway's handler of
    that command accepts (its `need(nrhs == ...)` guard in matlab/aps_mex.cpp), and against the number of outputs it fills.
"""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MATLAB = os.path.join(ROOT, "matlab")
TABLE = os.path.join(ROOT, "tests", "golden", "reference_signatures.json")
REF = "/root/reference/Procedural Program"

# mex entry points have no .m `function` line in the reference: their surface is the call site
MEX_CALL_SITES = {
    "flann_knn_win": ("featureMatching/featureMatchingGlobal.m", r"\[(\w+)\s*,\s*(\w+)\]\s*=\s*flann_knn_win\(([^)]*)\)"),
    "nearest2HammingExhaustiveMEX": ("featureMatching/matchFeaturesScratch.m", r"\[(\w+)\s*,\s*(\w+)\s*,\s*(\w+)\]\s*=\s*nearest2HammingExhaustiveMEX\(([^)]*)\)"),
    "nearest2HammingExhaustiveOMPMEX": ("featureMatching/matchFeaturesScratch.m", r"\[(\w+)\s*,\s*(\w+)\s*,\s*(\w+)\]\s*=\s*nearest2HammingExhaustiveOMPMEX\(([^)]*)\)"),
}


def parse_function_line(text):
    """(outs, name, ins) of the first `function` line of an .m file."""
    joined = re.sub(r"\.\.\.[^\n]*\n", " ", text)
    m = re.search(r"^\s*function\s+(?:\[([^\]]*)\]\s*=\s*|(\w+)\s*=\s*)?(\w+)\s*(?:\(([^)]*)\))?", joined, re.M)
    assert m, "no function line"
    outs = [o for o in re.split(r"[\s,]+", (m.group(1) or m.group(2) or "").strip()) if o]
    ins = [a for a in re.split(r"[\s,]+", (m.group(4) or "").strip()) if a]
    return outs, m.group(3), ins


def derive_table_from_reference():
    table = {}
    names = {f[:-2] for f in os.listdir(MATLAB) if f.endswith(".m")}
    for dirpath, _, files in os.walk(REF):
        for f in files:
            if f.endswith(".m") and f[:-2] in names:
                path = os.path.join(dirpath, f)
                outs, name, ins = parse_function_line(open(path, errors="replace").read())
                assert name == f[:-2]
                table[name] = {"outs": outs, "ins": ins, "ref": os.path.relpath(path, REF) + ":1"}
    for name, (rel, pat) in MEX_CALL_SITES.items():
        src = open(os.path.join(REF, rel), errors="replace").read()
        src = re.sub(r"\.\.\.[^\n]*\n", " ", src)
        m = re.search(pat, src)
        assert m, name
        line = src[:m.start()].count("\n") + 1
        nout = len(m.groups()) - 1
        args = [a.strip() for a in m.groups()[-1].split(",")]
        table[name] = {"outs": ["out%d" % (i + 1) for i in range(nout)], "ins": ["arg%d" % (i + 1) for i in range(len(args))],
                       "call_site": True, "ref": "%s:~%d" % (rel, line)}
    return table


def load_table():
    return json.load(open(TABLE))


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only in the build container")
def test_committed_table_equals_the_reference_function_lines():
    derived = derive_table_from_reference()
    committed = load_table()
    strip = lambda t: {k: {"outs": v["outs"], "ins": v["ins"]} for k, v in t.items()}
    assert strip(derived) == strip(committed), "regenerate with: python tests/test_shadow_signatures.py"


def test_every_shadow_declares_the_reference_signature():
    table = load_table()
    checked = 0
    for f in sorted(os.listdir(MATLAB)):
        if not f.endswith(".m") or f[:-2] not in table:
            continue
        outs, name, ins = parse_function_line(open(os.path.join(MATLAB, f)).read())
        want = table[name]
        if want.get("call_site"):
            # mex surface: the shadow must accept at least the call site's argument count and fill its outputs
            assert len(outs) >= len(want["outs"]), (f, outs, want["outs"])
            assert "varargin" in ins or len(ins) >= len(want["ins"]), (f, ins, want["ins"])
            fixed = [a for a in ins if a != "varargin"]
            assert len(fixed) <= len(want["ins"]), (f, "requires more arguments than %s passes" % want["ref"])
        else:
            assert outs == want["outs"], "%s: outputs %s, the reference (%s) declares %s" % (f, outs, want["ref"], want["outs"])
            assert ins == want["ins"], "%s: inputs %s, the reference (%s) declares %s" % (f, ins, want["ref"], want["ins"])
        checked += 1
    assert checked >= 14, checked   # the operator table of SURVEY 8(b)
    assert "imageMatching" in table and table["imageMatching"]["ins"][2:4] == ["keypoints", "matchesAll"]


def test_forwarding_calls_keep_the_argument_order():
    """aps_call_shadowed('<name>', mfilename('fullpath'), a, b, c ...) hands the reference's own file the shadow's
    arguments: they must be the declared inputs, in the declared order."""
    for f in sorted(os.listdir(MATLAB)):
        if not f.endswith(".m"):
            continue
        text = re.sub(r"\.\.\.[^\n]*\n", " ", open(os.path.join(MATLAB, f)).read())
        _, name, ins = parse_function_line(text)
        for m in re.finditer(r"aps_call_shadowed\('(\w+)',\s*mfilename\('fullpath'\)\s*,?([^;]*)\);", text):
            assert m.group(1) == name, (f, m.group(1))
            passed = [a.strip() for a in split_args(m.group(2))]
            if passed and passed[-1] == "varargin{:}":
                passed[-1] = "varargin"
            assert passed == ins[:len(passed)] and len(passed) >= len([a for a in ins if a != "varargin"]) - 1, (f, passed, ins)


def test_render_shadow_forwards_the_annotation_call_to_the_reference():
    """rgbAnnotation is filled only when opts.showPanoramaImgsNums && opts.showCropBoundingBox (renderPanorama.m:438-477,
    653-679; displayPanorama.m:126-136 stores it): the shadow must hand exactly that case, with both outputs and all seven
    inputs, to the reference's own file, and must do so BEFORE any device work."""
    text = re.sub(r"\.\.\.[^\n]*\n", " ", open(os.path.join(MATLAB, "renderPanorama.m")).read())
    text = "\n".join(l for l in text.split("\n") if not l.lstrip().startswith("%"))
    fwd = text.index("aps_call_shadowed('renderPanorama'")
    guard = text.rfind("if ", 0, fwd)
    cond = text[guard:fwd]
    assert "showPanoramaImgsNums" in cond and "showCropBoundingBox" in cond and "isfield" in cond
    assert re.search(r"\[panorama,\s*rgbAnnotation\]\s*=\s*aps_call_shadowed\('renderPanorama',\s*mfilename\('fullpath'\),\s*input,\s*images,\s*"
                     r"imgSize,\s*cameras,\s*mode,\s*refIdx,\s*opts\);\s*return", text)
    assert fwd < text.index("aps_mex(") and fwd < text.index("fillDefaults(opts")


def split_args(s):
    """Top-level comma split of a MATLAB argument list (strings, (), [], {} respected)."""
    out, depth, cur, i, in_str = [], 0, "", 0, False
    while i < len(s):
        c = s[i]
        if in_str:
            cur += c
            if c == "'":
                if i + 1 < len(s) and s[i + 1] == "'":
                    cur += "'"
                    i += 1
                else:
                    in_str = False
        elif c == "'" and (not cur.strip() or cur.rstrip()[-1] in "(,[{=+-*/<>~&| "):
            in_str = True
            cur += c
        elif c in "([{":
            depth += 1
            cur += c
        elif c in ")]}":
            depth -= 1
            cur += c
        elif c == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += c
        i += 1
    if cur.strip():
        out.append(cur)
    return out


def gateway_arity():
    """{command: (set of accepted nrhs or ('>=', n), max outputs)} read from matlab/aps_mex.cpp."""
    src = open(os.path.join(MATLAB, "aps_mex.cpp")).read()
    handlers = {}
    for m in re.finditer(r"static void (cmd_\w+)\(([^)]*)\)\s*\{(.*?)\n\}", src, re.S):
        body = m.group(3)
        k = body.find("nrhs")
        assert k >= 0, m.group(1)
        cond = body[k:body.find(";", k)].split("&&")[0].split(",")[0]      # the first guard on nrhs of the handler
        eq = [int(x) for x in re.findall(r"nrhs\s*==\s*(\d+)", cond)]
        ge = re.search(r"nrhs\s*>=\s*(\d+)", cond)
        nout = 1 + max([int(x) for x in re.findall(r"nlhs\s*>\s*(\d+)", body)] or [0])
        handlers[m.group(1)] = (set(eq) if eq else (">=", int(ge.group(1))), nout)
    table = {}
    for m in re.finditer(r'cmd == "(\w+)"\)\s*(cmd_\w+)\(', src):
        table[m.group(1)] = handlers[m.group(2)]
    return table


def test_every_aps_mex_call_has_an_arity_the_gateway_accepts():
    arity = gateway_arity()
    assert len(arity) >= 18, sorted(arity)
    seen = 0
    for f in sorted(os.listdir(MATLAB)):
        if not f.endswith(".m"):
            continue
        text = re.sub(r"\.\.\.[^\n]*\n", " ", open(os.path.join(MATLAB, f)).read())
        text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("%"))
        for m in re.finditer(r"(?:\[([^\]=]*)\]\s*=\s*|(\w+)\s*=\s*)?aps_mex\(", text):
            # balanced scan for the closing parenthesis
            j, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(text[j], 0)
                j += 1
            args = split_args(text[m.end():j - 1])
            cmd = args[0].strip().strip("'")
            if cmd not in arity:
                continue                      # 'version' / 'set_device': no handler function
            accepted, max_out = arity[cmd]
            nrhs = len(args)
            if any(a.strip().endswith("{:}") for a in args):
                continue                      # cell expansion: count is a run-time value
            if isinstance(accepted, set):
                assert nrhs in accepted, "%s: aps_mex('%s') with nrhs = %d, the gateway accepts %s" % (f, cmd, nrhs, sorted(accepted))
            else:
                assert nrhs >= accepted[1], (f, cmd, nrhs, accepted)
            nlhs = len([o for o in re.split(r"[\s,]+", (m.group(1) or "").strip()) if o]) if m.group(1) is not None else 1
            assert nlhs <= max_out, "%s: aps_mex('%s') asks for %d outputs, the gateway fills %d" % (f, cmd, nlhs, max_out)
            seen += 1
    assert seen >= 15, seen


if __name__ == "__main__":
    json.dump(derive_table_from_reference(), open(TABLE, "w"), indent=1, sort_keys=True)
    print("wrote", TABLE)
