"""The drop-in translation units (host/src/feature_match_mvo.cpp, g2o_ba_mvo.cpp) define everything the reference headers
declare for the two replaced sources (feature_match.h:12-54, g2o_ba.h:16-30), this repo's OpenCV-less mirror headers declare
the same functions, and host/tests/test_callsites.cpp calls every one of them -- a function that goes missing fails the build
of that program (built by __graft_entry__.build()) instead of a maintainer's link step.  tests/test_dropin_headers.py compiles
the translation units against the reference's own headers."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "monocular-visual-odometry_amd", "host")
# the declarations of the two replaced reference headers (names only; kept in step with /root/reference below)
FEATURE_MATCH_H = ["calcKeyPoints", "calcDescriptors", "matchFeatures", "matchByRadiusAndBruteForce", "removeDuplicatedMatches",
                   "selectUniformKptsByGrid", "computeMeanDistBetweenKeypoints", "inliers2DMatches", "pts2Keypts"]
G2O_BA_H = ["optimizeSingleFrame", "bundleAdjustment"]


def _declared(path):
    src = open(path).read()
    src = re.sub(r"//[^\n]*|/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:[A-Za-z_:<>\s\*&]+?)\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", src, flags=re.M)))


def test_name_lists_match_the_reference_headers_when_present():
    ref = "/root/reference/include/my_slam"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not on this machine (GPU box)")
    assert _declared(os.path.join(ref, "geometry", "feature_match.h")) == sorted(FEATURE_MATCH_H)
    assert _declared(os.path.join(ref, "optimization", "g2o_ba.h")) == sorted(G2O_BA_H)


def test_translation_units_define_and_mirror_headers_declare_every_function():
    fm = open(os.path.join(HOST, "src", "feature_match_mvo.cpp")).read()
    ba = open(os.path.join(HOST, "src", "g2o_ba_mvo.cpp")).read()
    for name in FEATURE_MATCH_H:
        assert re.search(r"^[A-Za-z_:<>\s\*&]*\b%s\s*\([^;{]*\)\s*\{" % name, fm, flags=re.M), "feature_match_mvo.cpp lacks %s" % name
    for name in G2O_BA_H:
        assert re.search(r"^[A-Za-z_:<>\s\*&]*\b%s\s*\([^;{]*\)\s*\{" % name, ba, flags=re.M), "g2o_ba_mvo.cpp lacks %s" % name
    # the mirror headers are declaration lists like the reference's (no definitions that could drift from the translation units)
    mfm = open(os.path.join(HOST, "include", "my_slam", "geometry", "feature_match.h")).read()
    mba = open(os.path.join(HOST, "include", "my_slam", "optimization", "g2o_ba.h")).read()
    for name, hdr in [(n, mfm) for n in FEATURE_MATCH_H] + [(n, mba) for n in G2O_BA_H]:
        code = re.sub(r"//[^\n]*|/\*.*?\*/", "", hdr, flags=re.S)
        assert not re.search(r"\b%s\s*\([^;{]*\)\s*(?:const\s*)?\{" % name, code), "mirror header DEFINES %s (must only declare it)" % name
        assert not re.search(r"\binline\b[^;{]*\b%s\s*\(" % name, code), "mirror header has an inline %s" % name
    for name in FEATURE_MATCH_H:
        assert re.search(r"\b%s\s*\([^;{]*\)\s*;" % name, mfm), "mirror feature_match.h does not declare %s" % name
    for name in G2O_BA_H:
        assert re.search(r"\b%s\s*\([^;{]*\)\s*;" % name, mba), "mirror g2o_ba.h does not declare %s" % name


def test_callsite_program_uses_every_function_and_links():
    src = open(os.path.join(HOST, "tests", "test_callsites.cpp")).read()
    frame = open(os.path.join(HOST, "include", "my_slam", "vo", "frame.h")).read()
    for name in FEATURE_MATCH_H:
        assert re.search(r"geometry::%s\s*\(" % name, src + frame), "no call site for geometry::%s" % name
    for name in G2O_BA_H:
        assert re.search(r"optimization::%s\s*\(" % name, src), "no call site for optimization::%s" % name
    exe = os.path.join(HOST, "tests", "test_callsites")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libmvo_hip.so" in ldd and "liboracle" not in ldd


def test_orb_parameters_are_latched_per_ctx():
    """ADVICE r1 + r2: hot_path_ctx() is thread_local and a thread may bind several ctxs in turn, so the 'parameters
    latched' flag and the pyramid-owner token are kept per (thread, ctx)."""
    fm = open(os.path.join(HOST, "src", "feature_match_mvo.cpp")).read()
    # (keyed by the ctx's unique id -- a ctx created at the address of a destroyed one must not inherit the flag)
    assert re.search(r"static\s+thread_local\s+std::unordered_map<unsigned long long,\s*CtxState>", fm)
    assert "m[mvo_ctx_uid(hot_path_ctx())]" in fm
