"""Socket power and core clock of the GPU this process drives, from the amdgpu driver's per-device hwmon files
(/sys/class/drm/card*/device/hwmon/hwmon*/{power1_input [uW], freq1_input [Hz], power1_cap [uW]}).  Read-only, unprivileged, optional:
every function returns None where the files are missing.  Used by bench.py (the line reports what the socket drew during the timed
steps: the GEMM launches of the step run AT the power cap) and tools/power_probe.py."""
import glob
import os
import threading
import time
from typing import Optional

import torch


def find_hwmon(device_index: Optional[int] = None) -> Optional[str]:
    """hwmon directory of a visible GPU, matched by the PCI address torch reports for it."""
    try:
        idx = torch.cuda.current_device() if device_index is None else device_index
        pr = torch.cuda.get_device_properties(idx)
        want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except (AttributeError, RuntimeError, AssertionError):
        return None
    for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        try:
            if f"PCI_SLOT_NAME={want}" in open(os.path.join(h, "device", "uevent")).read() and os.path.exists(os.path.join(h, "power1_input")):
                return h
        except OSError:
            continue
    return None


def power_cap_w(hw: Optional[str]) -> Optional[float]:
    try:
        return int(open(os.path.join(hw, "power1_cap")).read()) / 1e6
    except (OSError, ValueError, TypeError):
        return None


class Sampler(threading.Thread):
    """Samples (time, watts, sclk MHz) every `period` seconds until stop() — a host thread reading two small files."""

    def __init__(self, hw: str, period: float = 0.02):
        super().__init__(daemon=True)
        self.hw, self.period, self._stop_flag, self.rows = hw, period, False, []

    def run(self):
        while not self._stop_flag:
            try:
                p = int(open(os.path.join(self.hw, "power1_input")).read()) / 1e6
                f = int(open(os.path.join(self.hw, "freq1_input")).read()) / 1e6
                self.rows.append((time.perf_counter(), p, f))
            except (OSError, ValueError):
                pass
            time.sleep(self.period)

    def stop(self):
        self._stop_flag = True
        self.join()
        return self.rows

    def summary(self, since: float = 0.0) -> Optional[dict]:
        rows = [r for r in self.rows if r[0] >= since]
        if not rows:
            return None
        return {"mean_w": round(sum(r[1] for r in rows) / len(rows), 1), "max_w": round(max(r[1] for r in rows), 1),
                "sclk_mhz": round(sum(r[2] for r in rows) / len(rows)), "samples": len(rows)}
