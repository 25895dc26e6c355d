"""Build-time audit of the one-step LSTM layer kernel's device code (no GPU needed: hipcc
cross-compiles).  The kernel pins its upper-layer weights to accumulation registers with an asm
constraint; when the register allocator evicts such a vector it saves the pieces apart and, with
this compiler, does not restore all of them (tools/exp/orphan_spills.py describes the pattern).
The shipped objects must be free of it, and the checker must still find it in the form that is
known to have it.  The pinned builds are also assembled from a listing in which every packed
vector instruction that is directly followed by a 16-bit matrix instruction has been separated
from it by a wait state (tools/exp/packed_before_mfma.py has the case that asked for it)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kalman_vae_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CHECKER = os.path.join(ROOT, "tools", "exp", "orphan_spills.py")
PACKED = os.path.join(ROOT, "tools", "exp", "packed_before_mfma.py")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-S", "--cuda-device-only"]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")


def listing(tmp_path, name, extra):
    out = str(tmp_path / (name + ".s"))
    subprocess.run([HIPCC, *FLAGS, *extra, "-o", out, os.path.join(CSRC, "lstm_step_fused.hip")],
                   check=True, stderr=subprocess.DEVNULL)
    return out


def check(path, *args):
    return subprocess.run([sys.executable, CHECKER, path, *args], capture_output=True, text=True)


def test_upper_layer_object_has_no_spill(tmp_path):
    """The instances with pinned weights (compiled as the Makefile compiles them)."""
    r = check(listing(tmp_path, "upper", ["-DKVAE_SF_UPPER", "-mllvm", "-amdgpu-mfma-vgpr-form"]), "--any")
    assert r.returncode == 0, r.stdout


def test_pinned_lower_layer_object_has_no_spill(tmp_path):
    """The lower layer's two-input instances, pinned like the upper layer's."""
    r = check(listing(tmp_path, "lower2", ["-DKVAE_SF_LOWER2", "-mllvm", "-amdgpu-mfma-vgpr-form"]), "--any")
    assert r.returncode == 0, r.stdout


def test_packed_pairs_of_the_pinned_lower_layer_are_found_and_separated(tmp_path):
    """The listing the compiler gives for the lower layer's two-input instances has such pairs
    (the one that returned wrong values among them); the listing the build assembles has none,
    and differs from the compiler's by the wait states only."""
    raw = listing(tmp_path, "lower2", ["-DKVAE_SF_LOWER2", "-mllvm", "-amdgpu-mfma-vgpr-form"])
    r = subprocess.run([sys.executable, PACKED, raw], capture_output=True, text=True)
    assert r.returncode == 1 and "v_pk_" in r.stdout, r.stdout
    padded = str(tmp_path / "lower2.padded.s")
    subprocess.run([sys.executable, PACKED, raw, "-o", padded], check=True, capture_output=True)
    r = subprocess.run([sys.executable, PACKED, padded], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    a, b = open(raw).read().splitlines(), open(padded).read().splitlines()
    assert [x for x in b if x.strip() != "s_nop 0"] == [x for x in a if x.strip() != "s_nop 0"]
    assert len(b) > len(a)


def test_makefile_builds_the_pinned_units_from_separated_listings():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    padded = mk.split("PADDED_UNITS :=")[1].splitlines()[0].split()
    for unit in ("lstm_step_fused", "lstm_step_fused_upper", "lstm_step_fused_lower2", "rowstream_matmul",
                 "wgrad_tn", "lstm_resident_split", "lstm_resident_tiles_split"):
        assert unit in padded
    assert "NO_SPILL_UNITS := lstm_step_fused_upper lstm_step_fused_lower2" in mk


def test_lower_layer_object_restores_what_it_spills(tmp_path):
    r = check(listing(tmp_path, "lower", []))
    assert r.returncode == 0, r.stdout


def test_checker_finds_the_unrestored_piece_in_the_pinned_lower_layer():
    """The reproducer: the compiler's listing of the lower layer's one-input instance with its
    weights pinned in the default build (accumulators in accumulation registers as well), kept
    as a fixture from the day it returned wrong values: one dword of an evicted weight vector
    is saved to a register that nothing reads again.  (Whether the allocator evicts a pinned
    vector there changes with every edit of the kernel - today's source does not - which is
    why the build refuses any spill in a pinned object instead of looking for this pattern
    alone.)"""
    r = check(os.path.join(ROOT, "tests", "golden", "lstm_step_pinned_lower_gfx950.s"))
    assert r.returncode == 1 and "never read afterwards" in r.stdout, r.stdout
