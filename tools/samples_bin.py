#!/usr/bin/env python3
"""Reader of the binary sample sink of the C host layer (APEMOST_DUMP=binary -> samples.bin):
a 64-byte header ("APEMOSTB", u32 version = 2, n_beta, n_par, n_swap, u64 thin, u32 n_param_chains), then
per kept iteration the parameter vectors of chains 0..n_param_chains-1 followed by (prob, prob - prior)
of every chain, all doubles -- what the reference's text files hold.

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
    """-> (header dict, params [iterations][n_param_chains][n_par], probs [iterations][n_beta][2])"""
    with open(path, "rb") as f:
        head = f.read(64)
    if head[:8] != b"APEMOSTB":
        raise SystemExit("%s: not a samples.bin file" % path)
    version, n_beta, n_par, n_swap = struct.unpack("<4I", head[8:24])
    thin, = struct.unpack("<Q", head[24:32])
    n_pc, = struct.unpack("<I", head[32:36])
    if version != 2:
        raise SystemExit("%s: format version %d, this reader knows 2" % (path, version))
    record = n_pc * n_par + 2 * n_beta
    raw = np.memmap(path, dtype=np.float64, mode="r", offset=64)
    raw = raw[: len(raw) // record * record].reshape(-1, record)
    hdr = dict(version=version, n_beta=n_beta, n_par=n_par, n_swap=n_swap, thin=thin, n_param_chains=n_pc)
    return hdr, raw[:, : n_pc * n_par].reshape(-1, n_pc, n_par), raw[:, n_pc * n_par:].reshape(-1, n_beta, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--params", help="the params file (parameter names for the per-parameter dumps)")
    ap.add_argument("--text", help="directory to write the reference's text dumps into")
    a = ap.parse_args()
    hdr, params, probs = read(a.file)
    print("%s: %d iterations, (prob, prob-prior) of %d chains, %d parameters of chains 0..%d, thin %d, n_swap %d" %
          (a.file, len(probs), hdr["n_beta"], hdr["n_par"], hdr["n_param_chains"] - 1, hdr["thin"], hdr["n_swap"]))
    if not a.text:
        return
    names = ["p%d" % i for i in range(hdr["n_par"])]
    if a.params:
        names = [line.split()[3] for line in open(a.params) if line.strip()]
    os.makedirs(a.text, exist_ok=True)
    for c in range(hdr["n_param_chains"]):
        for p in range(hdr["n_par"]):
            with open(os.path.join(a.text, "%s-chain-%d.prob.dump" % (names[p], c)), "w") as f:
                f.write("".join("%.15e\n" % v for v in params[:, c, p]))
    for c in range(hdr["n_beta"]):
        with open(os.path.join(a.text, "prob-chain%d.dump" % c), "w") as f:
            f.write("".join("%6e\t%6e\n" % (v[0], v[1]) for v in probs[:, c]))


if __name__ == "__main__":
    main()
