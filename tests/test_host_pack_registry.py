"""CPU: PackRegistry's census (networks/elements.py): fragment buffers only the plan-time trials read stop being refreshed,
come back when somebody asks for them, and retiring them invalidates recorded tapes (version bump)."""
import torch

from nlt_amd.networks.elements import PackRegistry


class FakeLayer:
    def __init__(self):
        self.kernel = torch.zeros(4)
        self._packed = {}
        self.ver = 0

    def _version(self):
        return self.ver


def test_census_retires_unused_buffers_and_reactivates_on_demand(monkeypatch):
    from nlt_amd import _capi as C
    tables = []
    monkeypatch.setattr(C, 'repack_table', lambda rows, dev: (tables.append([r['dst'] for r in rows]) or (len(rows), len(rows), 1)))
    monkeypatch.setattr(C, 'repack_weights', lambda *a: None)
    state = [0]
    reg = PackRegistry(lambda: state[0])
    layers = [FakeLayer() for _ in range(3)]
    bufs = {}
    for i, l in enumerate(layers):
        for key in ('a', 'b'):
            bufs[(i, key)] = torch.zeros(2)
            l._packed[key] = (0, bufs[(i, key)])
            reg.add(l, key, bufs[(i, key)], {})
    v0 = reg.version
    reg.refresh()
    assert len(tables[-1]) == 6
    reg.begin_census(passes=2)
    for _ in range(2):                              # two plan passes that only ever ask for the 'a' layouts
        reg.tick()
        for l in layers:
            reg.touch(l, 'a')
    assert not reg.inactive and reg.version == v0
    reg.tick()                                      # third pass begins: the census closes
    assert reg.used is None and reg.version == v0 + 1            # tapes recorded before are invalid now
    assert reg.inactive == {(id(l), 'b') for l in layers}
    state[0] = 1
    for l in layers:
        l.ver = 1
    reg.refresh_if_stale()
    assert len(tables[-1]) == 3 and all(any(t is bufs[(i, 'a')] for t in tables[-1]) for i in range(3))
    assert all(l._packed['a'][0] == 1 and l._packed['b'][0] == 0 for l in layers)        # 'b' stays stale ...
    reg.touch(layers[1], 'b')                       # (no census: nothing recorded)
    reg.activate(layers[1], 'b')                    # ... until somebody asks (what _cached_pack does on a stale owned entry)
    reg.refresh()
    assert len(tables[-1]) == 4 and layers[1]._packed['b'][0] == 1 and layers[0]._packed['b'][0] == 0
    reg.prune()                                     # no census running: a no-op
    assert reg.version == v0 + 1
    reg.drop(layers[0])
    assert (id(layers[0]), 'b') not in reg.inactive and len(reg.entries) == 4
