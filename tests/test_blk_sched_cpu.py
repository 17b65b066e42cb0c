"""How demod_blk_kernel's calls are cut into time slices (webaudio_modem_amd/csrc/fsk_blk_sched.h): the header is plain
integer arithmetic, compiled here with g++ into a tiny program that evaluates it on a grid of batch sizes and call
lengths.  CPU only."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

SRC = r'''
#include <cstdio>
#include <cstdlib>
#include "fsk_blk_sched.h"
int main(int argc, char **argv) {
  // groups n_tiles resident slice_tiles have_queue  ->  ns st
  for (int i = 1; i + 4 < argc; i += 5) {
    uint32_t st = 12345;
    const uint32_t ns = fsk::blk_slice_count((uint32_t)strtoul(argv[i], 0, 10), (uint32_t)strtoul(argv[i + 1], 0, 10),
                                             (uint32_t)strtoul(argv[i + 2], 0, 10), (uint32_t)strtoul(argv[i + 3], 0, 10),
                                             atoi(argv[i + 4]) != 0, &st);
    printf("%u %u\n", ns, st);
  }
  return 0;
}
'''


@pytest.fixture(scope="module")
def sched(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("g++ not installed")
    d = tmp_path_factory.mktemp("blk_sched")
    (d / "t.cc").write_text(SRC)
    exe = str(d / "t")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "webaudio_modem_amd", "csrc"), "-o", exe, str(d / "t.cc")],
                   check=True)

    def run(cases):
        args = [str(v) for c in cases for v in c]
        out = subprocess.run([exe] + args, check=True, capture_output=True, text=True).stdout.split()
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(len(cases))]
    return run


OFF = 0xFFFFFFFF


def test_not_sliced_within_one_round_or_without_a_queue_or_when_off(sched):
    r = sched([(1024, 30000, 1024, 0, 1), (64, 30000, 1024, 0, 1), (1280, 30000, 1024, 0, 0), (1280, 30000, 1024, OFF, 1),
               (1280, 30000, 0, 0, 1), (1280, 700, 1024, 0, 1), (1280, 768, 1024, 0, 1)])
    assert r == [(1, 0)] * 7


def test_slices_cover_the_call_and_respect_the_bounds(sched):
    cases = [(g, n, 1024, 0, 1) for g in (1025, 1088, 1280, 1536, 2047, 2048, 4096, 8192, 100000)
             for n in (769, 1500, 6000, 30000, 100000, 1000000)]
    for (g, n, *_), (ns, st) in zip(cases, sched(cases)):
        assert 2 <= ns <= 128, (g, n, ns, st)
        assert st * ns >= n > st * (ns - 1), (g, n, ns, st)            # whole call, last slice not empty
        assert st >= 96 or st >= -(-n // 128), (g, n, ns, st)          # no slivers (unless the 128-slice cap forces them)


def test_slice_count_is_chosen_for_packing(sched):
    # 1 088 groups on 1 024 resident workgroups, 6 000 tiles: 8 slices of 750 would need 9 slice times (1.125 rounds); 16
    # need 17 (1.0625)
    (ns, st), = sched([(1088, 6000, 1024, 0, 1)])
    assert ns >= 15 and -(-1088 * ns // 1024) / ns < 1.07, (ns, st)
    # 1 280 groups: 8 slices are 10 slice times = 1.25 rounds exactly; more slices only add slice changes
    (ns, st), = sched([(1280, 6000, 1024, 0, 1)])
    assert ns == 8 and st == 750, (ns, st)
    # two full rounds: nothing to gain either
    (ns, st), = sched([(2048, 6000, 1024, 0, 1)])
    assert ns == 8, (ns, st)


def test_an_explicit_slice_length_is_taken_as_given(sched):
    assert sched([(16, 100, 5, 7, 1), (16, 100, 5, 1, 1), (16, 1000, 5, 3, 1)]) == [(15, 7), (100, 1), (125, 8)]
