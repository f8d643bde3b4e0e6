"""Synthetic input generators for the DFA-scan workloads of BASELINE.json / SURVEY.md section 8(d).

Pure data generation (numpy; an equivalent torch form for on-device generation): no matching
logic lives here.  Every byte is a function of (seed, absolute byte index) through a
counter-based SplitMix64, so a shard can be generated anywhere - on the host for the CPU
checker, on the GPU for the timed run - and be identical.
"""
from __future__ import annotations

import numpy as np

ALPHABET47 = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789 ./:-_=&?%@", dtype=np.uint8)
assert len(ALPHABET47) == 47

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(idx: np.ndarray, seed: int) -> np.ndarray:
    """SplitMix64 output number idx (0-based) of the stream seeded with `seed`."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def random_bytes(n: int, seed: int, start: int = 0) -> np.ndarray:
    """n uniform bytes 0..255: byte k of the stream is byte (k & 7) of SplitMix64 word k >> 3."""
    first, last = start >> 3, (start + n + 7) >> 3
    words = splitmix64(np.arange(first, last, dtype=np.uint64), seed)
    return words.view(np.uint8)[start - (first << 3): start - (first << 3) + n].copy()


def alphabet_bytes(n: int, seed: int, start: int = 0) -> np.ndarray:
    """n bytes drawn from the 47-character log/URL alphabet."""
    return ALPHABET47[random_bytes(n, seed, start) % 47]


def fixed_lines(n_lines: int, line_len: int, seed: int, *, alphabet: bool = True,
                plant: bytes | None = None, plant_every: int = 8, plant_at: int = 32,
                first_line: int = 0) -> np.ndarray:
    """n_lines x line_len bytes, contiguous (stride == line_len).  `plant` is written at
    byte `plant_at` of every `plant_every`-th line (global line number)."""
    start = first_line * line_len
    gen = alphabet_bytes if alphabet else random_bytes
    buf = gen(n_lines * line_len, seed, start).reshape(n_lines, line_len)
    if plant:
        p = np.frombuffer(plant, dtype=np.uint8)[: max(0, line_len - plant_at)]
        rows = np.arange(n_lines)[(np.arange(n_lines) + first_line) % plant_every == 0]
        buf[rows[:, None], plant_at + np.arange(len(p))[None, :]] = p[None, :]
    return buf.reshape(-1)


def ragged_lines(n_lines: int, min_len: int, max_len: int, seed: int, *, alphabet: bool = True,
                 heads: list[bytes] | None = None, head_every: int = 2):
    """Variable-length lines: (data uint8[total], offsets uint64[n_lines+1]).  Every
    `head_every`-th line starts with heads[i % len(heads)] (a planted signature instance)."""
    lens = (splitmix64(np.arange(n_lines, dtype=np.uint64), seed ^ 0x5EED) %
            np.uint64(max_len - min_len + 1)).astype(np.int64) + min_len
    offsets = np.zeros(n_lines + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens).astype(np.uint64)
    total = int(offsets[-1])
    data = (alphabet_bytes if alphabet else random_bytes)(total, seed)
    if heads:
        for i in range(0, n_lines, head_every):
            h = np.frombuffer(heads[(i // head_every) % len(heads)], dtype=np.uint8)
            k = min(len(h), int(lens[i]))
            o = int(offsets[i])
            data[o:o + k] = h[:k]
    return data, offsets


# ---- the reference-compiled config DFAs' pattern sets ----------------
URI_REGEX = (r"(https?|ftps?|file|wss?|sftp|ssh|git|ldaps?)://"
             r"(([a-z0-9-]+\.)+[a-z]{2,6}|[0-9]{1,3}\.[0-9]{1,3}\.[0-9]{1,3}\.[0-9]{1,3})"
             r"(:[0-9]{1,5})?(/[A-Za-z0-9._~%!$&'()*+,;=:@/-]*)?"
             r"(\?[A-Za-z0-9._~%!$&'()*+,;=:@/?-]*)?(#[A-Za-z0-9._~%-]*)?")
URI_PLANT = b"https://ab-c.example.com:8080/p/x.y?q=1&r=%20#frag "
# SURVEY 8a's "userinfo variant": 343 states / 25 classes / 18,124 B through the reference
URI_USER_REGEX = URI_REGEX.replace("://(", "://([a-z0-9._-]+@)?(", 1)
URI_USER_PLANT = b"ssh://deploy.bot@build-7.example.org:2222/~/repo.git "

# BASELINE.json configs[4]'s real-regex stand-in (SURVEY 8d "C5" alternative): 20 schemes,
# userinfo, names / IPv4 / bracketed IPv6, loose start, ignore case -> 3,254 states and 35
# classes through the reference compiler: a class table of 228 KB that does NOT fit LDS whole.
_H16 = r"[0-9a-f]{1,4}"
URI_V6_REGEX = (
    r"(https?|ftps?|file|wss?|sftp|ssh|git|ldaps?|mailto|news|nntp|telnet|gopher|irc|rtsp|smb|"
    r"nfs|svn|s3|gs)://([a-z0-9._-]+@)?"
    r"(([a-z0-9-]+\.)+[a-z]{2,6}|[0-9]{1,3}\.[0-9]{1,3}\.[0-9]{1,3}\.[0-9]{1,3}|"
    r"\[((H:){7}H|(H:){1,6}:H|::(H:){0,5}H)\])"
    r"(:[0-9]{1,5})?(/[A-Za-z0-9._~%!$&'()*+,;=:@/-]*)?"
    r"(\?[A-Za-z0-9._~%!$&'()*+,;=:@/?-]*)?(#[A-Za-z0-9._~%-]*)?").replace("H", _H16)
URI_V6_PLANT = b"SFTP://bob@[2001:db8:0:1:0:0:0:2f]:2222/srv/data?x=1#top "

LOG_LEVELS = ["ERROR", "WARN", "INFO", "DEBUG", "FATAL"]
LOG_SUBSYS = ["net", "disk", "auth", "db", "cache", "sched", "rpc", "dns", "tls", "fs",
              "mem", "cpu", "gpu", "raid", "ntp", "smtp", "http", "kern", "init", "cron"]


def log100_patterns():
    """100 log-signature regexes -> results 1..100 (BASELINE.json config 4)."""
    pats = []
    for li, lvl in enumerate(LOG_LEVELS):
        for si, sub in enumerate(LOG_SUBSYS):
            rx = lvl + r" \[" + sub + r"\] [a-z]+ (failed|timeout|refused|ok)( code=[0-9]+)?"
            pats.append((rx, 1 + li * len(LOG_SUBSYS) + si, 0))
    return pats


def log100_heads():
    verbs = [b"failed", b"timeout", b"refused", b"ok"]
    heads = []
    k = 0
    for lvl in LOG_LEVELS:
        for sub in LOG_SUBSYS:
            h = lvl.encode() + b" [" + sub.encode() + b"] " + ALPHABET47[:3 + k % 5].tobytes() + \
                b" " + verbs[k % 4] + (b" code=%d" % (k * 7) if k % 3 == 0 else b"") + b" "
            heads.append(h)
            k += 1
    return heads
