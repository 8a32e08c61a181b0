"""The GPU's current shader clock from sysfs (amdgpu hwmon `freq1_input`, label sclk), for measurement code only: bench.py samples it
from a side thread while a kernel runs back to back and prices the issue ceiling at THAT clock next to the 2.4 GHz one
(the reference has nothing comparable).  No GPU API is involved; returns None where the file is absent."""
import glob
import os

_cache = {}


def _cards():
    """sysfs device directories of the amdgpu cards, in PCI bus order (the order HIP enumerates them in on these boxes)."""
    out = []
    for d in glob.glob("/sys/class/drm/card[0-9]*/device"):
        if os.path.basename(os.path.dirname(d)).count("-"):
            continue
        try:
            if "amdgpu" not in os.path.realpath(os.path.join(d, "driver")):
                continue
        except OSError:
            continue
        out.append((os.path.basename(os.path.realpath(d)), d))
    return [d for _, d in sorted(out)]


def source(device=0, pci=None):
    """Path of the file read_mhz() reads for HIP device `device` (or the card at PCI address `pci`), or None."""
    key = (device, pci)
    if key in _cache:
        return _cache[key]
    cards = _cards()
    if pci:
        cards = [d for d in cards if os.path.basename(os.path.realpath(d)).lower() == pci.lower()]
        device = 0
    path = None
    if device < len(cards):
        for f in sorted(glob.glob(os.path.join(cards[device], "hwmon", "hwmon*", "freq*_input"))):
            try:
                label = open(f.replace("_input", "_label")).read().strip()
            except OSError:
                label = "sclk" if f.endswith("freq1_input") else ""
            if label == "sclk":
                path = f
                break
    _cache[key] = path
    return path


def read_mhz(device=0, pci=None):
    p = source(device, pci)
    if not p:
        return None
    try:
        return int(open(p).read()) / 1e6
    except (OSError, ValueError):
        return None
