"""Structural checks on the gfx950 code of the fused merge kernel (no GPU: hipcc cross-compiles to assembly).

The whole-wave path of k_tile_sums issues its loads as inline assembly and waits for them with hand-counted
`s_waitcnt vmcnt(N)` (kmdiff_amd/csrc/kmd_tilemerge.hip, fetch_w / vm_wait): that is only right while nothing the
compiler adds -- a register spilled to scratch, say -- is in flight between them.  Held here for every instantiation."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def tile_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "tile.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "kmdiff_amd", "csrc", "kmd_tilemerge.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernels(lines):
    name, body = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\S*k_tile_sums\S+):", l)
        if m:
            name = m.group(1)
            body[name] = []
        elif ".amdhsa_kernel" in l:
            name = None
        elif name:
            body[name].append(l)
    return body


@pytest.mark.timeout(300)
def test_nothing_of_the_compilers_in_flight_between_the_hand_counted_loads(tile_asm):
    ks = kernels(tile_asm)
    # (the hand-issued loads carry the comment "; ring")
    wide = {k: b for k, b in ks.items() if any("buffer_load_dword" in l and "; ring" in l for l in b)}
    assert len(wide) == 16, sorted(ks)            # (2 shapes) x (filter / rows) x (one / two limbs) x (32- / 64-bit sums), whole waves
    for k, b in wide.items():
        at = [i for i, l in enumerate(b) if "buffer_load_dword" in l and "; ring" in l]
        span = b[at[0]:at[-1] + 1]
        assert not [l for l in span if "scratch_" in l], k
        # compiler-issued vector memory operations inside the span: none (its own loads would be counted by vmcnt too)
        assert not [l for l in span if re.search(r"\b(global|flat)_(load|store|atomic)", l)], k
        two_limbs = re.search(r"ELb[01]ELb1ELb1ELb[01]EEEv", k) is not None      # <threads, slots, filter, TWO LIMBS, whole waves, 32-bit sums>
        loads_per_round = 3 if two_limbs else 2
        assert len(at) == loads_per_round * 8, (k, len(at))      # 4 rounds in the prologue + 4 ring stages
        waits = [l for l in span if "s_waitcnt vmcnt" in l]
        want = "s_waitcnt vmcnt(%d)" % (loads_per_round * 3)
        assert waits and all(want in l for l in waits), (k, waits[:4], want)


def test_buckets_are_read_with_one_16_byte_lds_read(tile_asm):
    ks = kernels(tile_asm)
    hot = [b for k, b in ks.items() if "ILi512ELj2048ELb1ELb0ELb1ELb1E" in k]
    assert len(hot) == 1
    # the hand-written stage 1: per ring stage once in the middle of a run and once for its last round
    assert sum("ds_read_b128 v[40:43]" in l for l in hot[0]) == 8
