"""Host logic around HIP-graph capture (closerlook3d_amd/fused.py): the whole_step_capture declaration and the
deferred CSR join of the gather operators.  No GPU: only the bookkeeping is exercised."""
import torch

import closerlook3d_amd
from closerlook3d_amd import fused


def test_whole_step_capture_is_scoped_and_restores_the_previous_state():
    assert fused._WHOLE_STEP[0] is False
    with closerlook3d_amd.whole_step_capture():
        assert fused._WHOLE_STEP[0] is True
        with closerlook3d_amd.whole_step_capture(False):
            assert fused._WHOLE_STEP[0] is False
        assert fused._WHOLE_STEP[0] is True
    assert fused._WHOLE_STEP[0] is False
    try:
        with closerlook3d_amd.whole_step_capture():
            raise RuntimeError("capture failed")
    except RuntimeError:
        pass
    assert fused._WHOLE_STEP[0] is False  # restored when the capture raises


def test_summary_switch_follows_the_module_flag(monkeypatch):
    # (outside a capture the stream is not capturing: the summary is used whenever the switch is on)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(fused, "SUPPORT_SUMMARY", True)
    assert fused._use_summary()
    monkeypatch.setattr(fused, "SUPPORT_SUMMARY", False)
    assert not fused._use_summary()


def test_an_undeclared_capture_takes_the_slot_walk(monkeypatch):
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    monkeypatch.setattr(fused, "SUPPORT_SUMMARY", True)
    assert not fused._use_summary()
    with closerlook3d_amd.whole_step_capture():
        assert fused._use_summary()


def test_deferred_join_marks_the_output_once(monkeypatch):
    joined = []
    monkeypatch.setattr(fused, "_join_inverse", lambda idx: joined.append(idx))
    out, idx = torch.zeros(2), torch.zeros(3, dtype=torch.int32)
    assert fused._deferred(out, idx, False) is out and not hasattr(out, "_cl3d_pending")
    fused.join_pending(out)  # nothing pending: no join
    assert joined == []
    fused._deferred(out, idx, True)
    fused.join_pending(out)
    fused.join_pending(out)  # the second call finds nothing
    assert len(joined) == 1 and joined[0] is idx
