"""The MATLAB-side binding (matlab/) never meets MATLAB in this image; this is what CAN be checked on a CPU:
  * the mex gateway is type-correct C++ against the documented MEX API (a declarations-only tests/mex_decl/mex.h) and
    against include/aps.h - every C-ABI call it makes has the right arity and types;
  * every command string a shadowing .m wrapper sends exists in the gateway's dispatcher;
  * every .m file is structurally sound (balanced function/end blocks, a single main function named like the file) and
    shadows a function that really exists under the same name in the reference tree layout listed in INTEGRATION.md."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MATLAB = os.path.join(ROOT, "matlab")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_mex_gateway_is_type_correct_against_the_mex_api_and_aps_h():
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "tests", "mex_decl"), os.path.join(MATLAB, "aps_mex.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _gateway_commands():
    src = open(os.path.join(MATLAB, "aps_mex.cpp")).read()
    return set(re.findall(r'cmd == "([a-z0-9_]+)"', src))


def test_every_wrapper_command_exists_in_the_gateway():
    cmds = _gateway_commands()
    assert {"sift_extract", "match_features", "render", "image_warp", "crop_nonzero_bbox", "multiband_blend"} <= cmds
    used = set()
    for f in os.listdir(MATLAB):
        if f.endswith(".m"):
            used |= set(re.findall(r"aps_mex\('([a-z0-9_]+)'", open(os.path.join(MATLAB, f)).read()))
    assert used and used <= cmds, used - cmds


def _strip(code):
    out = []
    for line in code.splitlines():
        line = re.sub(r"'[^']*'", "''", line)       # string literals
        line = line.split("%")[0]                    # comments
        out.append(line)
    return "\n".join(out)


def test_m_files_are_structurally_sound():
    openers = re.compile(r"\b(function|if|for|while|switch|try|parfor|arguments)\b")
    for f in sorted(os.listdir(MATLAB)):
        if not f.endswith(".m"):
            continue
        code = _strip(open(os.path.join(MATLAB, f)).read())
        m = re.search(r"^\s*function\s+(?:\[[^\]]*\]\s*=\s*|\w+\s*=\s*)?(\w+)", code, re.M)
        assert m and m.group(1) == f[:-2], f"{f}: main function must be named like the file"
        n_open = len(openers.findall(code))
        n_end = len(re.findall(r"\bend\b", re.sub(r"\([^()]*\bend\b[^()]*\)", "()", code)))  # x(end) is an index, not a block end
        assert n_open == n_end, f"{f}: {n_open} block openers vs {n_end} ends"
        assert code.count("(") == code.count(")") and code.count("[") == code.count("]"), f


def test_shadows_cover_the_operator_table_of_survey_8b():
    have = {f[:-2] for f in os.listdir(MATLAB) if f.endswith(".m")}
    need = {"getFeaturePoints", "featureMatchingPairwise", "matchFeaturesScratch", "flann_knn_win",
            "nearest2HammingExhaustiveMEX", "nearest2HammingExhaustiveOMPMEX", "estimateTransformationRANSAC",
            "estimateTransformationMLESAC", "renderPanorama", "multiBandBlending", "linearBlending", "imageWarp",
            "featureMatchingGlobal", "imageMatching"}
    assert need <= have, need - have
