"""The measured-and-dropped variants of the one-barrier kernel stay in the source behind compile-time switches
(APEMOST_OB_WAVE_PERM, APEMOST_OWNER_PRIO_PHASE, APEMOST_OB_FLAG_LATE, APEMOST_EXP_OWNER_SLACK,
APEMOST_OB_HELPER_SIMD, APEMOST_PHILOX_MERGED, APEMOST_MERGED_ALL, APEMOST_EXP_NO_ATTEMPTS, APEMOST_STAMP_PHASES).  Code that is
never compiled rots: one development build (pulse model, four likelihood waves: with and without the helper wavefront)
with the switches that change the most code, cross-compiled for gfx950 -- no GPU needed; their parity and rates were
measured on the box (profiles/r04_wave_perm_and_c5_coop.txt, r04_owner_prio_phase.txt, r04_owner_slack_c2.txt)."""
import os

from apemost_amd import build


def test_the_switched_off_variants_still_compile(tmp_path):
    out = build.build_dev([1], [4], out=str(tmp_path / "switches.so"),
                          extra=["-DAPEMOST_OWNER_PRIO_PHASE=1", "-DAPEMOST_OB_FLAG_LATE=1", "-DAPEMOST_EXP_OWNER_SLACK=1",
                                 "-DAPEMOST_PHILOX_MERGED=1", "-DAPEMOST_STAMPS", "-DAPEMOST_STAMP_PHASES"])
    assert os.path.getsize(out) > 100000
    out = build.build_dev([1], [4], out=str(tmp_path / "switches2.so"),
                          extra=["-DAPEMOST_OB_WAVE_PERM=2", "-DAPEMOST_OB_PART_FIRST=0", "-DAPEMOST_OB_SKIP_LOOPS=0", "-DAPEMOST_OB_HELPER_SIMD=1",
                                 "-DAPEMOST_OB_HELPER_EU=4", "-DAPEMOST_EXP_NO_ATTEMPTS=1", "-DAPEMOST_PHILOX_MERGED=0", "-DAPEMOST_MERGED_ALL=1"])
    assert os.path.getsize(out) > 100000
