"""The config-5 harness (tools/config5_dryrun.py, BASELINE.json configs[4]) - its arithmetic and log parsing only; the encodes
themselves are a build-container tool run (profiles/r03_config5_dryrun.txt), not part of the suite."""
import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_bd_rate_known_answers():
    import config5_dryrun as C5
    rate = np.array([4000.0, 2200.0, 1200.0, 650.0]); psnr = np.array([42.0, 39.5, 37.0, 34.4])
    assert abs(C5.bd_rate(rate, psnr, rate, psnr)) < 1e-9
    assert abs(C5.bd_rate(rate, psnr, 0.9 * rate, psnr) + 10.0) < 1e-6          # 10 % fewer bits at every quality
    assert abs(C5.bd_rate(rate, psnr, 1.25 * rate, psnr) - 25.0) < 1e-6
    worse = C5.bd_rate(rate, psnr, rate, psnr - 0.3)                             # same bits, lower quality -> positive
    assert 0 < worse < 20


def test_encoder_log_parser():
    import config5_dryrun as C5
    log = ("POC    0 TId: 0 ( I-SLICE, QP 22 )\n\nSUMMARY --------------------------------------------------------\n"
           "\tTotal Frames |   Bitrate     Y-PSNR    U-PSNR    V-PSNR    YUV-PSNR \n"
           "\t        2    a    1790.4000   41.9472   43.5538   44.2128   42.4356\n\n"
           "finished @ Sat Oct  3 12:00:00 2026\n Total Time:       12.345 sec. [user]       12.400 sec. [elapsed]\n")
    kbps, y, yuv, secs = C5.parse_encoder_log(log)
    assert (kbps, y, yuv, secs) == (1790.4, 41.9472, 42.4356, 12.345)
