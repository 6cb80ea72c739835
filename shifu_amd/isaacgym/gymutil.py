"""`gymutil.parse_device_str` (shifu/gym/isaac_gym.py:215)."""


def parse_device_str(device_str):
    s = str(device_str).lower()
    if s in ("cpu",):
        return "cpu", 0
    if s in ("cuda", "gpu"):
        return "cuda", 0
    if s.startswith("cuda:") or s.startswith("gpu:"):
        return "cuda", int(s.split(":")[1])
    raise ValueError(f"invalid device string {device_str!r}")
