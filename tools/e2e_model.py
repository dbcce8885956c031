#!/usr/bin/env python3
"""Timeline model of decode(host array): what any ordering of the trellis work on ONE stream can reach.

Inputs are measured on one MI355X box in round 6 (N=128, M=16, D=39, T=1000; ms):
  * upload: 1024 utterances of 312 KB in 16 chunks, 0.3475 ms per chunk (57.6 GB/s, page-locked source);
  * emission of one 64-utterance chunk: slice_features 0.018 + emit_diag_i8 0.281 = 0.30 (whole batch in one launch: 4.70);
  * Viterbi kernel by batch size (viterbi_reg1b_n128_u1 up to 384 utterances, the pair kernel from 512): the frame chain is
    latency below one block per CU, so a quarter of the batch costs more than a third of the whole;
  * back-trace 0.26 whatever the batch (one dependent chain of T steps per utterance), paths back 0.08.

The GPU is the longer pipeline (6.1 ms of kernels against 5.56 ms of upload), its kernels cannot share a CU with the emission
kernel (256 registers x 2 waves per SIMD: the whole file), so every schedule is a permutation of the same serial work plus
what splitting the Viterbi launch costs.  Result: nothing beats "all emission, then one trellis" by more than 0.05 ms; the
floor of the scheme is first chunk + serial GPU work = 0.35 + 4.80 + 1.38 + 0.26 + 0.08 = 6.87 ms (1.13 x the resident step),
and what separates the measured 7.6-7.8 ms from it are the 16 gaps of 0.05 ms in which the emission stream waits for the
next chunk -- gaps no kernel of this path fits into.

    python tools/e2e_model.py
"""
V = {64: 0.446, 128: 0.470, 192: 0.496, 256: 0.510, 320: 0.731, 384: 0.803, 512: 0.982, 768: 1.345, 1024: 1.378}
UP, EMIT, BT, D2H, PER = 0.3475, 0.30, 0.26, 0.08, 64


def end_time(groups):
    """groups: utterances per trellis launch, in order; each launch follows the emission of its own chunks"""
    t, c = 0.0, 0
    for g in groups:
        for _ in range(g // PER):
            t = max(t, UP * (c + 1)) + EMIT          # the chunk's copy must have arrived
            c += 1
        t += V[g]
    return t + BT + D2H


if __name__ == "__main__":
    resident = 0.265 + 4.433 + 1.378 + 0.265
    for name, g in (("all emission, then one trellis (shipped)", [1024]), ("halves", [512, 512]), ("quarters", [256] * 4),
                    ("768 + 256", [768, 256]), ("256 + 768", [256, 768]), ("512 + 256 + 256", [512, 256, 256])):
        e = end_time(g)
        print("%-42s %.2f ms   %.3f x max(upload, resident step)" % (name, e, e / max(16 * UP, resident)))
    print("floor: first chunk + serial GPU work                %.2f ms" % (UP + 16 * EMIT + V[1024] + BT + D2H))
