"""Clocks / power / temperature of the AMD GPUs of this host from sysfs, sampled by a thread while
a timed region runs (bench.py, tools/bench_c5.py: `config.gpu_state`).  Files only -- no rocm-smi
child process is started from a process that uses the GPU.  Box-to-box spreads of a bench line
(25.9e3 against 30.8e3 evals/s) can then be read against the clocks the box ran at.

A box may show more cards in sysfs than HIP sees; every amdgpu card is listed, the busiest one
(highest mean power over the region) first -- that is the one the region ran on."""
import glob
import os
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _current_mhz(text):
    """'0: 132Mhz\\n1: 2100Mhz *\\n' -> 2100.0 (the starred level)."""
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith('*'):
            try:
                return float(line.split(':')[1].lower().replace('mhz', '').replace('*', ''))
            except (IndexError, ValueError):
                return None
    return None


def cards():
    out = []
    for dev in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
        if (_read(os.path.join(dev, 'vendor')) or '').strip() != '0x1002':
            continue
        hw = sorted(glob.glob(os.path.join(dev, 'hwmon', 'hwmon*')))
        out.append((dev, hw[0] if hw else None))
    return out


def sample(card):
    dev, hw = card
    s = {'sclk_mhz': _current_mhz(_read(os.path.join(dev, 'pp_dpm_sclk'))),
         'mclk_mhz': _current_mhz(_read(os.path.join(dev, 'pp_dpm_mclk'))),
         'fclk_mhz': _current_mhz(_read(os.path.join(dev, 'pp_dpm_fclk'))),
         'socclk_mhz': _current_mhz(_read(os.path.join(dev, 'pp_dpm_socclk')))}
    if hw:
        for key, name, scale in (('power_w', 'power1_average', 1e-6),
                                 ('power_w', 'power1_input', 1e-6),
                                 ('power_cap_w', 'power1_cap', 1e-6),
                                 ('temp_c', 'temp1_input', 1e-3),
                                 ('temp_c', 'temp2_input', 1e-3)):
            if s.get(key) is None:
                v = _read(os.path.join(hw, name))
                try:
                    s[key] = float(v) * scale if v else None
                except ValueError:
                    s[key] = None
    busy = _read(os.path.join(dev, 'gpu_busy_percent'))
    try:
        s['busy_pct'] = float(busy) if busy else None
    except ValueError:
        s['busy_pct'] = None
    return s


def current_pci():
    """'dddd:bb:dd' of torch's current HIP device (None when torch cannot say): the sysfs card
    whose PCI address starts with it is the one this process runs on."""
    try:
        import torch
        p = torch.cuda.get_device_properties(torch.cuda.current_device())
        return f'{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}'
    except Exception:                                            # noqa: BLE001
        return None


class Sampler:
    """with Sampler() as st: <timed region>;  st.summary() -> dict or None.  pci: PCI address
    prefix of the card to report (default: torch's current device when it can be identified in
    sysfs, else every card, the one drawing most power first)."""

    def __init__(self, period=0.05, pci='auto'):
        self.period = period
        self.cards = [] if os.environ.get('PB_GPU_STATE') == '0' else cards()
        self.pci = current_pci() if pci == 'auto' else pci
        if self.pci:
            mine = [c for c in self.cards
                    if os.path.basename(os.path.realpath(c[0])).startswith(self.pci)]
            if mine:
                self.cards = mine
            else:
                self.pci = None
        self.rows = [[] for _ in self.cards]
        self._stop = threading.Event()
        self._thread = None

    def __enter__(self):
        # (re-usable: make the object BEFORE the chip is warmed up -- finding the cards in sysfs
        # takes milliseconds, and milliseconds of idle between the priming spectra and a short
        # timed region cost 7 % of bench.py's 20-step value -- and enter it right at the region)
        self.rows = [[] for _ in self.cards]
        self._stop = threading.Event()
        if self.cards:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def _run(self):
        while not self._stop.is_set():
            for rows, card in zip(self.rows, self.cards):
                rows.append(sample(card))
            self._stop.wait(self.period)

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2)
        return False

    def summary(self):
        def stat(rows, key):
            v = [r[key] for r in rows if r.get(key) is not None]
            if not v:
                return None
            return {'mean': round(sum(v) / len(v), 1), 'min': min(v), 'max': max(v)}
        out = []
        for (dev, _), rows in zip(self.cards, self.rows):
            if not rows:
                continue
            out.append({'card': os.path.basename(os.path.dirname(dev)), 'samples': len(rows),
                        **{k: stat(rows, k) for k in ('sclk_mhz', 'mclk_mhz', 'fclk_mhz',
                                                      'socclk_mhz', 'power_w', 'power_cap_w',
                                                      'temp_c', 'busy_pct')}})
        if not out:
            return None
        out.sort(key=lambda c: -((c['power_w'] or {}).get('mean') or 0.0))
        return {'source': 'sysfs (amdgpu pp_dpm_*, hwmon)', 'pci': self.pci, 'cards': out[:2],
                'identified': bool(self.pci)}


if __name__ == '__main__':
    with Sampler() as st:
        time.sleep(0.3)
    print(st.summary())
