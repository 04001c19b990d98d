"""Clock and power of one GPU, sampled from sysfs by a host thread (bench.py's `multi_gpu` / `sensors` objects).

The dominant kernels of this path sit on a power / clock ridge (0.72 MFMA-busy at 1.67 GHz under the package's
power cap), so the first question about a sub-linear 8-GPU curve is "did the clocks drop when eight sockets ran at once?".  This module
answers it without touching the launch path: plain file reads of the amdgpu hwmon nodes, nothing from the HIP runtime, no subprocess.

    /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input     sclk, Hz
    /sys/class/drm/card*/device/hwmon/hwmon*/power1_average   socket power, microwatts (power1_input on parts that lack the average)
    /sys/class/drm/card*/device/pp_dpm_sclk                   fallback for the clock: the level marked '*'

A card is matched to a torch device by PCI bus id when torch exposes one, else by position among the amdgpu cards.
Everything is best-effort: a node that cannot be read yields None fields, never an exception.
"""
import glob
import os
import re
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def amdgpu_cards():
    """[(card_dir, pci_bus_id)] of the amdgpu devices, sorted by PCI address (the order HIP enumerates them in by default)."""
    out = []
    for d in glob.glob("/sys/class/drm/card[0-9]*/device"):
        if _read(os.path.join(d, "vendor")) != "0x1002":
            continue
        real = os.path.realpath(d)
        out.append((d, os.path.basename(real)))
    out.sort(key=lambda t: t[1])
    return out


def card_for_device(index, pci_bus_id=None):
    cards = amdgpu_cards()
    if pci_bus_id:
        want = pci_bus_id.lower()
        for d, bus in cards:
            if bus.lower().endswith(want[-7:]) or want.endswith(bus.lower()[-7:]):
                return d
    return cards[index][0] if 0 <= index < len(cards) else None


def read_once(card_dir):
    """{"sclk_mhz", "power_w"} of one card right now (None where unreadable)."""
    sclk = power = None
    if card_dir:
        for hw in glob.glob(os.path.join(card_dir, "hwmon", "hwmon*")):
            v = _read(os.path.join(hw, "freq1_input"))
            if v and v.isdigit():
                sclk = int(v) / 1e6
            for node in ("power1_average", "power1_input"):
                p = _read(os.path.join(hw, node))
                if p and p.isdigit():
                    power = int(p) / 1e6
                    break
        if sclk is None:
            txt = _read(os.path.join(card_dir, "pp_dpm_sclk")) or ""
            m = re.search(r"(\d+)\s*[Mm]hz\s*\*", txt)
            if m:
                sclk = float(m.group(1))
    return {"sclk_mhz": sclk, "power_w": power}


class Sampler:
    """with Sampler(card_dir, period_s=0.02) as s: ...timed region...; s.summary() -> mean / min / max of clock and power."""

    def __init__(self, card_dir, period_s=0.02):
        self.card_dir, self.period = card_dir, period_s
        self.samples = []
        self._stop = threading.Event()
        self._th = None

    def _run(self):
        while not self._stop.is_set():
            self.samples.append(read_once(self.card_dir))
            self._stop.wait(self.period)

    def __enter__(self):
        if self.card_dir:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th:
            self._th.join(timeout=1.0)
        return False

    def summary(self):
        def stat(key):
            v = [s[key] for s in self.samples if s.get(key) is not None]
            if not v:
                return None
            return {"mean": round(sum(v) / len(v), 1), "min": round(min(v), 1), "max": round(max(v), 1)}
        return {"samples": len(self.samples), "period_ms": round(self.period * 1e3, 1), "sclk_mhz": stat("sclk_mhz"),
                "power_w": stat("power_w"), "source": ("sysfs hwmon of " + self.card_dir) if self.card_dir else None}


def for_torch_device(index):
    """card directory of torch's cuda:<index> (None if it cannot be found)."""
    bus = None
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        if hasattr(p, "pci_bus_id"):
            bus = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
    except Exception:      # noqa: BLE001 - best effort
        bus = None
    return card_for_device(index, bus)


if __name__ == "__main__":
    for d, bus in amdgpu_cards():
        print(d, bus, read_once(d))
