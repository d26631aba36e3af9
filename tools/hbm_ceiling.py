"""Practical HBM ceilings of the box, for the roofline discussion: device-to-device copy
(1 read + 1 write per byte), fill (write only) and a reduction (read only), 1 GiB buffers, via torch."""
import torch


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    b = torch.empty(n, dtype=torch.uint8, device="cuda")
    f = torch.empty(n // 4, dtype=torch.float32, device="cuda")
    t = timed(lambda: b.copy_(a))
    print(f"copy  1 GiB -> 1 GiB : {t * 1e6:8.1f} us  {2 * n / t / 1e12:5.2f} TB/s moved ({n / t / 1e12:5.2f} TB/s each way)")
    t = timed(lambda: a.fill_(7))
    print(f"fill  1 GiB          : {t * 1e6:8.1f} us  {n / t / 1e12:5.2f} TB/s written")
    t = timed(lambda: f.sum())
    print(f"sum   1 GiB (f32)    : {t * 1e6:8.1f} us  {n / t / 1e12:5.2f} TB/s read")


if __name__ == "__main__":
    main()
