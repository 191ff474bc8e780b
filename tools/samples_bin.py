#!/usr/bin/env python3
"""Reader of the binary sample sink of the C host layer (APEMOST_DUMP=binary -> samples.bin):
a 64-byte header ("APEMOSTB", u32 version, n_beta, n_par, n_swap, u64 thin), then per kept iteration
n_beta rows of n_par+2 doubles (params..., prob, prob - prior).

    python tools/samples_bin.py samples.bin                       # summary
    python tools/samples_bin.py samples.bin --params params --text DIR
        expands it into the reference's text files under DIR: <name>-chain-0.prob.dump ("%.15e",
        src/mcmc_dump.c:79-88) and prob-chain<i>.dump ("%6e\\t%6e", src/parallel_tempering.c:399-401),
        which `analyse` and the reference's tools read."""
import argparse
import os
import struct

import numpy as np


def read(path):
    with open(path, "rb") as f:
        head = f.read(64)
    if head[:8] != b"APEMOSTB":
        raise SystemExit("%s: not a samples.bin file" % path)
    version, n_beta, n_par, n_swap = struct.unpack("<4I", head[8:24])
    thin, = struct.unpack("<Q", head[24:32])
    rows = np.memmap(path, dtype=np.float64, mode="r", offset=64)
    rows = rows[: len(rows) // (n_beta * (n_par + 2)) * n_beta * (n_par + 2)].reshape(-1, n_beta, n_par + 2)
    return dict(version=version, n_beta=n_beta, n_par=n_par, n_swap=n_swap, thin=thin), rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--params", help="the params file (parameter names for the per-parameter dumps)")
    ap.add_argument("--text", help="directory to write the reference's text dumps into")
    ap.add_argument("--all-chains", action="store_true", help="per-parameter dumps for every chain (-DDUMP_ALL_CHAINS)")
    a = ap.parse_args()
    hdr, rows = read(a.file)
    print("%s: %d iterations x %d chains x (%d parameters + prob, prob-prior), thin %d, n_swap %d" %
          (a.file, len(rows), hdr["n_beta"], hdr["n_par"], hdr["thin"], hdr["n_swap"]))
    if not a.text:
        return
    names = ["p%d" % i for i in range(hdr["n_par"])]
    if a.params:
        names = [line.split()[3] for line in open(a.params) if line.strip()]
    os.makedirs(a.text, exist_ok=True)
    n_par = hdr["n_par"]
    for c in range(hdr["n_beta"] if a.all_chains else 1):
        for p in range(n_par):
            with open(os.path.join(a.text, "%s-chain-%d.prob.dump" % (names[p], c)), "w") as f:
                f.write("".join("%.15e\n" % v for v in rows[:, c, p]))
    for c in range(hdr["n_beta"]):
        with open(os.path.join(a.text, "prob-chain%d.dump" % c), "w") as f:
            f.write("".join("%6e\t%6e\n" % (v[0], v[1]) for v in rows[:, c, n_par:n_par + 2]))


if __name__ == "__main__":
    main()
