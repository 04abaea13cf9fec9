"""Deterministic synthetic "PNG-filter-like" byte buffers.

The reference defines its workloads only as distributions (benches/bench.rs:24-75, unseeded
`rand::thread_rng()`), so the build owns the PRNG: a counter-mode splitmix64.  Byte `j` of
stream `i` is a pure function of (seed_i, j), which lets the same bytes be produced by numpy
on the host (tests) and by torch on the device (bench, no PCIe traffic).

Models (benches/bench.rs):
  D  bench_distribution :61-75   (default)
  M  bench_mixture      :46-58
  L  bench_low          :35-43
  U  bench_uniform_random :24-32
"""
import numpy as np

GOLDEN = 0x9E3779B97F4A7C15
BASE_SEED = 0xF0DEF1A7E
ROW_BYTES = 1024
MASK64 = (1 << 64) - 1


def _mix_np(z):
    z = z.astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def splitmix64_scalar(x):
    """One splitmix64 output for state x (python ints)."""
    z = (x + GOLDEN) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def stream_seed(i):
    return splitmix64_scalar(BASE_SEED ^ int(i))


def _bytes_from_words_np(v, model):
    r = ((v >> np.uint64(40)) % np.uint64(100)).astype(np.int64)
    lo = (v & np.uint64(0xFF)).astype(np.int64)
    mid = ((v >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int64)
    if model == "D":
        out = np.zeros(v.shape, dtype=np.int64)
        out = np.where(r == 0, lo, out)
        out = np.where((r >= 1) & (r <= 2), (mid & 31) - 16, out)
        out = np.where((r >= 11) & (r <= 50), (mid & 15) - 8, out)
        out = np.where((r >= 51) & (r <= 80), (mid & 7) - 4, out)
    elif model == "M":
        r200 = ((v >> np.uint64(40)) % np.uint64(200)).astype(np.int64)
        out = np.where(r200 == 1, lo, (mid & 31) - 16)
    elif model == "L":
        out = (mid & 15) * 2 - 16
    elif model == "U":
        out = lo
    else:
        raise ValueError(model)
    return (out & 0xFF).astype(np.uint8)


def gen_stream_np(i, length=65536, model="D", png_rows=True):
    """Raw buffer `i` of the batch as a numpy uint8 array."""
    seed = stream_seed(i)
    with np.errstate(over="ignore"):
        ctr = np.uint64(seed) + (np.arange(1, length + 1, dtype=np.uint64) * np.uint64(GOLDEN))
        v = _mix_np(ctr)
    out = _bytes_from_words_np(v, model)
    if png_rows:
        out = _apply_rows_np(out, i, v)
    return out


def _apply_rows_np(out, i, v):
    n = out.size
    idx = np.arange(n)
    col = idx % ROW_BYTES
    # byte 0 of each scanline is the PNG filter type 0..4
    ft = ((v >> np.uint64(16)) % np.uint64(5)).astype(np.uint8)
    out = np.where(col == 0, ft, out)
    kind = int(i) % 16
    if kind == 15:
        out = np.zeros_like(out)
    elif kind == 7:
        row = idx // ROW_BYTES
        out = np.where(row % 2 == 1, np.uint8(0), out)
    return out.astype(np.uint8)


def gen_batch_np(first, count, length=65536, model="D", png_rows=True):
    return [gen_stream_np(first + k, length, model, png_rows) for k in range(count)]


# ---------------------------------------------------------------------------------------
# torch (device) implementation -- bit-identical to the numpy one
# ---------------------------------------------------------------------------------------

def _lsr(x, s):
    """logical shift right on int64 tensors"""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _wrap(c):
    """python int -> signed 64-bit constant"""
    c &= MASK64
    return c - (1 << 64) if c >= (1 << 63) else c


def gen_batch_torch(first, count, length=65536, model="D", png_rows=True, device="cuda",
                    chunk=512):
    """[count, length] uint8 tensor on `device`; same bytes as gen_stream_np."""
    import torch

    out = torch.empty((count, length), dtype=torch.uint8, device=device)
    j = torch.arange(1, length + 1, dtype=torch.int64, device=device) * _wrap(GOLDEN)
    idx = torch.arange(length, device=device)
    col = idx % ROW_BYTES
    row = idx // ROW_BYTES
    for c0 in range(0, count, chunk):
        c1 = min(count, c0 + chunk)
        seeds = torch.tensor([_wrap(stream_seed(first + k)) for k in range(c0, c1)],
                             dtype=torch.int64, device=device)
        z = seeds[:, None] + j[None, :]
        z = (z ^ _lsr(z, 30)) * _wrap(0xBF58476D1CE4E5B9)
        z = (z ^ _lsr(z, 27)) * _wrap(0x94D049BB133111EB)
        v = z ^ _lsr(z, 31)
        r = _lsr(v, 40) % 100
        lo = v & 0xFF
        mid = _lsr(v, 8) & 0xFF
        if model == "D":
            b = torch.zeros_like(v)
            b = torch.where(r == 0, lo, b)
            b = torch.where((r >= 1) & (r <= 2), (mid & 31) - 16, b)
            b = torch.where((r >= 11) & (r <= 50), (mid & 15) - 8, b)
            b = torch.where((r >= 51) & (r <= 80), (mid & 7) - 4, b)
        elif model == "M":
            r200 = _lsr(v, 40) % 200
            b = torch.where(r200 == 1, lo, (mid & 31) - 16)
        elif model == "L":
            b = (mid & 15) * 2 - 16
        elif model == "U":
            b = lo
        else:
            raise ValueError(model)
        b = b & 0xFF
        if png_rows:
            ft = _lsr(v, 16) % 5
            b = torch.where(col[None, :] == 0, ft, b)
            kind = (torch.arange(c0, c1, device=device) + first) % 16
            b = torch.where((kind == 15)[:, None], torch.zeros_like(b), b)
            half = (kind == 7)[:, None] & (row % 2 == 1)[None, :]
            b = torch.where(half, torch.zeros_like(b), b)
        out[c0:c1] = b.to(torch.uint8)
    return out
