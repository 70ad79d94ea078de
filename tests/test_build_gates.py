"""The two build gates of csrc/Makefile (no GPU needed): check_resources.py rejects scratch in the kernels that wait on hand-counted
vmcnt, check_loops.py rejects scratch inside the tap-reading loops of k_fwd_brick_groups and the accumulation loop of k_bwd_brick,
and refuses to pass vacuously."""
import os
import pytest
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multiviewhmr_amd", "csrc")


def _run(script, path, rules=("k_fwd_brick_groups:ds_read_b128",)):
    args = [path] if script == "check_resources.py" else list(rules) + ["--", path]
    return subprocess.run([sys.executable, os.path.join(CSRC, script)] + args, capture_output=True, text=True)


def test_check_resources_flags_scratch_in_the_counted_kernels(tmp_path):
    remark = "x.hip:1:1: remark: Function Name: %s [-Rpass-analysis=kernel-resource-usage]\nx.hip:1:1: remark:     ScratchSize [bytes/lane]: %d [-Rpass-analysis=kernel-resource-usage]\n"
    ok = tmp_path / "ok.txt"
    ok.write_text(remark % ("_ZN5mvhmr11k_fwd_brickILi0ELi4ELi1024EfLi2EEEv", 0) + remark % ("_ZN5mvhmr11k_bwd_brickILi0ELi4ELi1024EfLi16EEEv", 452)
                  + remark % ("_ZN5mvhmr18k_fwd_brick_groupsILi0ELi8EfEEv", 372))
    assert _run("check_resources.py", str(ok)).returncode == 0            # backward and groups: their loops are checked on the asm instead
    bad = tmp_path / "bad.txt"
    bad.write_text(remark % ("_ZN5mvhmr11k_fwd_brickILi0ELi4ELi1024EfLi2EEEv", 80))
    r = _run("check_resources.py", str(bad))
    assert r.returncode == 1 and "scratch" in r.stderr


def test_check_loops_flags_scratch_next_to_the_tap_reads(tmp_path):
    head = "_ZN5mvhmr18k_fwd_brick_groupsILi0ELi8EfEEvPK:\n"
    cold = ".LBB3_152:\n\tscratch_store_dword off, v1, off\n\tds_read_b128 v[0:3], v9\n"                      # not a loop block
    loop_ok = ".LBB3_232:                              ;   in Loop: Header=BB3_232 Depth=1\n\tds_read_b128 v[0:3], v9\n\tv_mul_f32_e32 v0, v0, v7\n"
    loop_bad = ".LBB3_244:                              ;   in Loop: Header=BB3_232 Depth=1\n\tds_read_b128 v[0:3], v9\n\tscratch_load_dword v7, off, off offset:40\n"
    other_loop = ".LBB3_300:                              ;   in Loop: Header=BB3_300 Depth=1\n\tscratch_load_dword v7, off, off\n\tglobal_load_dwordx4 v[0:3], v[4:5], off\n"
    tail = ".Lfunc_end3:\n"
    ok = tmp_path / "ok.s"
    ok.write_text(head + cold + loop_ok + other_loop + tail)               # the slow path's loop (no tap reads) may spill
    assert _run("check_loops.py", str(ok)).returncode == 0
    bad = tmp_path / "bad.s"
    bad.write_text(head + cold + loop_ok + loop_bad + tail)
    r = _run("check_loops.py", str(bad))
    assert r.returncode == 1 and "BB3_232" in r.stderr                      # a spill in ANY block of the loop that holds the tap reads
    # the spill sits in a latch block WITHOUT tap reads of the same loop: still that loop
    latch = ".LBB3_245:                              ;   in Loop: Header=BB3_232 Depth=1\n\tscratch_load_dword v7, off, off offset:40\n\ts_cbranch_scc0 .LBB3_232\n"
    split = tmp_path / "split.s"
    split.write_text(head + cold + loop_ok + latch + tail)
    assert _run("check_loops.py", str(split)).returncode == 1
    # a fall-through block (no label) inside the loop
    ft = "; %bb.246:                              ;   in Loop: Header=BB3_232 Depth=1\n\tscratch_store_dword off, v7, off\n"
    fall = tmp_path / "fall.s"
    fall.write_text(head + cold + loop_ok + ft + tail)
    assert _run("check_loops.py", str(fall)).returncode == 1


def test_check_loops_does_not_pass_vacuously(tmp_path):
    renamed = tmp_path / "renamed.s"
    renamed.write_text("_ZN5mvhmr9k_renamedILi0EEEvPK:\n.LBB0_1:                              ;   in Loop: Header=BB0_1 Depth=1\n\tds_read_b128 v[0:3], v9\n.Lfunc_end0:\n")
    r = _run("check_loops.py", str(renamed))
    assert r.returncode == 1 and "no function matching" in r.stderr
    no_loop = tmp_path / "noloop.s"
    no_loop.write_text("_ZN5mvhmr18k_fwd_brick_groupsILi0ELi8EfEEvPK:\n.LBB0_1:\n\tv_mul_f32_e32 v0, v0, v7\n.Lfunc_end0:\n")
    r = _run("check_loops.py", str(no_loop))
    assert r.returncode == 1 and "no loop with" in r.stderr


def test_check_loops_covers_the_backward_accumulation_loop(tmp_path):
    head = "_ZN5mvhmr11k_bwd_brickILi0ELi4ELi1024EfLi16EEEvPK:\n"
    loop = ".LBB7_10:                              ;   in Loop: Header=BB7_10 Depth=1\n\tds_add_u32 v3, v4\n"
    spill = ".LBB7_11:                              ;   in Loop: Header=BB7_10 Depth=1\n\tscratch_store_dword off, v7, off\n"
    ok, bad = tmp_path / "ok.s", tmp_path / "bad.s"
    ok.write_text(head + loop + ".Lfunc_end7:\n")
    bad.write_text(head + loop + spill + ".Lfunc_end7:\n")
    rules = ("k_bwd_brick:ds_add_u32",)
    assert _run("check_loops.py", str(ok), rules).returncode == 0
    assert _run("check_loops.py", str(bad), rules).returncode == 1


def test_plane_backward_has_no_global_atomics_in_its_isa(tmp_path):
    """the coarse-grid backward accumulates in LDS and writes plain stores.  Disassemble the unit for gfx950 (hipcc
    cross-compiles without a GPU) and look at its three kernel families: the tap table, the Jacobian pass (k_plane_ds: plain stores
    only) and the plane kernel (k_bwd_plane: 16 LDS adds per voxel, plain stores)."""
    asm = tmp_path / "plane.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
                           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only", "-o", str(asm),
                           os.path.join(CSRC, "unproject_plane_bwd.hip")], stderr=subprocess.DEVNULL)
    inside, planes, jacobians, adds = None, 0, 0, 0
    for line in asm.read_text().splitlines():
        ls = line.strip()
        if ls.startswith("_ZN5mvhmr") and ("k_bwd_plane" in ls or "k_plane_ds" in ls or "k_plane_taps" in ls) and ls.endswith(":") is False and ":" in ls:
            inside = ls.split(":")[0]
            planes += "k_bwd_plane" in inside
            jacobians += "k_plane_ds" in inside
        elif ls.startswith(".Lfunc_end"):
            inside = None
        elif inside and not ls.startswith(";"):
            assert "global_atomic" not in ls and "buffer_atomic" not in ls and "flat_atomic" not in ls, (inside, ls)
            adds += "k_bwd_plane" in inside and (ls.startswith("ds_add_u32") or ls.startswith("ds_add_u64"))
    # fp32 / fp16 gradient storage x (4-byte cells with the two-walk 64-bit form inside | 8-byte cells | 8-byte cells, 4 private images); 16 LDS adds per voxel
    assert planes == 6 and adds >= 16 * planes
    assert jacobians >= 4 * 3 * 3                                                  # 4 methods x 3 view counts x 3 grad_out storage types


def test_the_library_was_built_from_these_sources():
    """a failed `make` leaves the previous libmvhmr_unproject.so in place and every test would then run against old kernels (it
    happened in round 4: a resource check failed one unit, the link step never ran, the failure sat behind a grep).  The link step
    stamps the library with a hash of its sources; the tree must still hash to it (independent of file times: the GPU box gets a copy)."""
    import glob, hashlib
    lib = os.path.join(ROOT, "multiviewhmr_amd", "lib")
    stamp = os.path.join(lib, "sources.md5")
    if not os.path.exists(os.path.join(lib, "libmvhmr_unproject.so")) or not os.path.exists(stamp):
        pytest.skip("no built library (or one from before the stamp) here")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "mvhmr_unproject.h")])
    h = hashlib.md5()
    for f in srcs:
        h.update(open(f, "rb").read())
    assert h.hexdigest() == open(stamp).read().strip(), \
        "multiviewhmr_amd/lib/libmvhmr_unproject.so was not built from the sources in the tree: run `make -C multiviewhmr_amd/csrc` and read its exit status"
