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


class _FakeEvent:
    pass


def _idx_with_pending_build():
    idx = torch.zeros(3, dtype=torch.int32)
    idx._cl3d_inverse = (7, torch.zeros(1), torch.zeros(1), _FakeEvent())
    return idx


def test_forward_end_join_is_skipped_only_in_a_declared_capture(monkeypatch):
    """_join_geometry: eager -> joined; captured without the declaration -> joined (the capture may end with the forward
    pass); captured under whole_step_capture() -> left to the backward's support-major pass and remembered."""
    joined = []
    monkeypatch.setattr(fused, "_join_inverse", lambda idx: joined.append(idx))
    for capturing, declared, expect_join in ((False, False, True), (False, True, True), (True, False, True),
                                             (True, True, False)):
        monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda c=capturing: c)
        del joined[:]
        idx = _idx_with_pending_build()
        if declared:
            try:
                with closerlook3d_amd.whole_step_capture():
                    fused._join_geometry(idx)
                    assert (len(joined) == 1) == expect_join
                    assert (idx in fused._PENDING) == (not expect_join)
                    del fused._PENDING[:]  # (what the backward's inverse_index() does when it joins)
            finally:
                pass
        else:
            fused._join_geometry(idx)
            assert (len(joined) == 1) == expect_join


def test_a_declared_capture_without_its_backward_is_reported(monkeypatch):
    """a forward captured under whole_step_capture() whose backward never joins the forked CSR build ends the
    capture unjoined; the context manager says so by name."""
    import pytest
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    monkeypatch.setattr(fused, "_join_inverse", lambda idx: None)
    with pytest.raises(RuntimeError, match="without their backward pass"):
        with closerlook3d_amd.whole_step_capture():
            fused._join_geometry(_idx_with_pending_build())
    assert fused._PENDING == [] and fused._WHOLE_STEP[0] is False
    with pytest.raises(RuntimeError, match="no backward pass joined it") as e:
        with closerlook3d_amd.whole_step_capture():
            fused._join_geometry(_idx_with_pending_build())
            raise ValueError("hipErrorStreamCaptureUnjoined")  # (what torch.cuda.graph's exit raises in that case)
    assert isinstance(e.value.__cause__, ValueError)


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
