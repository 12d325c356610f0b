"""The robot-side wire format of the reference's use_lcm path (robot_state_control_lcmt): BIT-EXACT against fixtures
produced by the reference's own generated codec (tests/golden/make_robot_state_golden.py)."""
import os

import numpy as np
import pytest

from oracle import traj_oracle as to

HERE = os.path.dirname(os.path.abspath(__file__))


def gold():
    raw = open(os.path.join(HERE, "golden", "robot_state_msgs.bin"), "rb").read()
    assert len(raw) % 204 == 0
    return [raw[i:i + 204] for i in range(0, len(raw), 204)], np.load(os.path.join(HERE, "golden", "robot_state_expected.npz"))


def test_oracle_codec_matches_the_reference_codec():
    msgs, exp = gold()
    for k, b in enumerate(msgs):
        q, v, tau = to.robot_state_decode(b)
        assert q.tobytes() == exp["q"][k].tobytes() and v.tobytes() == exp["v"][k].tobytes() and tau.tobytes() == exp["tau"][k].tobytes()
    order, act = list(exp["order"]), list(exp["act_joint"])
    ctl = exp["control_msgs"].tobytes()
    for i in range(exp["u"].shape[1]):
        assert to.robot_control_message(exp["u"][:, i], order, act) == ctl[204 * i:204 * i + 204]
    with pytest.raises(ValueError, match="Decode error"):
        to.robot_state_decode(b"\x00" * 204)


def test_c_codec_is_bit_exact_and_rejects_bad_input():
    from quadruped_drake_amd.lcm_io import decode_robot_state, encode_robot_state
    msgs, exp = gold()
    for k, b in enumerate(msgs):
        d = decode_robot_state(b)
        assert d["q"].astype(np.float64).tobytes() == exp["q"][k].tobytes()          # -0.0 and float32 denormals included
        assert d["v"].astype(np.float64).tobytes() == exp["v"][k].tobytes()
        assert d["tau"].astype(np.float64).tobytes() == exp["tau"][k].tobytes()
        assert encode_robot_state(d["q"], d["v"], d["tau"]) == b                     # round trip reproduces the wire bytes
    assert np.signbit(decode_robot_state(msgs[5])["q"][1])
    # double -> float32 rounds to nearest even, like the reference's struct.pack('>f')
    x = np.array([1.0 + 2.0 ** -24, 1.0 + 3 * 2.0 ** -24, 1.0 + 2.0 ** -23] + [0.0] * 15)
    assert decode_robot_state(encode_robot_state(v=x))["v"][:3].tolist() == [1.0, 1.0 + 2.0 ** -22, 1.0 + 2.0 ** -23]
    with pytest.raises(ValueError, match="Decode error"):
        decode_robot_state(b"\x00" * 8 + msgs[0][8:])
    with pytest.raises(ValueError):
        decode_robot_state(msgs[0][:203])


@pytest.mark.gpu
def test_device_unpack_and_pack_match_the_reference_codec():
    import torch
    from quadruped_drake_amd.lcm_io import pack_controls, unpack_states
    msgs, exp = gold()
    bad = bytearray(msgs[3]); bad[0] ^= 0xff                     # one foreign message in the batch
    batch = msgs[:3] + [bytes(bad)] + msgs[4:]
    q, v, ok = unpack_states(batch)
    q, v, ok = q.cpu().numpy(), v.cpu().numpy(), ok.cpu().numpy()
    assert ok.tolist() == [1, 1, 1, 0] + [1] * (len(msgs) - 4)
    keep = ok == 1
    assert q[:, keep].tobytes() == exp["q"][keep].T.copy().tobytes() and v[:, keep].tobytes() == exp["v"][keep].T.copy().tobytes()
    assert not q[:, 3].any() and not v[:, 3].any()               # untouched columns of the rejected message
    out = pack_controls(torch.tensor(exp["u"], device="cuda:0"), q_perm=exp["order"], act_perm=exp["act_joint"])
    assert out.cpu().numpy().tobytes() == exp["control_msgs"].tobytes()
    ident = pack_controls(torch.tensor(exp["u"], device="cuda:0")).cpu().numpy().tobytes()
    for i in range(exp["u"].shape[1]):
        assert ident[204 * i:204 * i + 204] == to.robot_control_message(exp["u"][:, i], list(range(12)), list(range(12)))
    with pytest.raises(Exception):
        pack_controls(torch.tensor(exp["u"], device="cuda:0"), q_perm=[0] * 12)


@pytest.mark.gpu
def test_wire_to_wire_tick_on_a_permuted_plant():
    """robot_current_state messages in, robot_control_input messages out, the tick in between -- the use_lcm loop of
    basic_controller.py:289-314 for a batch, on a plant with its own joint / actuator numbering, against the oracle fed
    the float32-rounded states the messages carry."""
    import torch
    from oracle import oracle_py as orc
    from quadruped_drake_amd import IDController, workloads
    from quadruped_drake_amd.lcm_io import encode_robot_state, pack_controls, unpack_states
    _, exp = gold()
    order, act = [int(x) for x in exp["order"]], [int(x) for x in exp["act_joint"]]
    b = workloads.make_batch(2, n=64)
    qd, vd = b["q"].copy(), b["v"].copy()
    qd[7 + np.array(order)] = b["q"][7:]; vd[6 + np.array(order)] = b["v"][6:]          # the plant's own joint order
    msgs = [encode_robot_state(qd[:, i], vd[:, i]) for i in range(64)]
    q, v, ok = unpack_states(msgs)
    assert ok.cpu().numpy().all()
    ctrl = IDController(model=b["model"], max_batch=64, device=0, q_perm=order, act_perm=act)
    up = lambda x: torch.tensor(np.ascontiguousarray(x), device="cuda:0")
    tau, met, st = ctrl.step(q, v, up(b["targets"]), up(b["mask"]))
    out = pack_controls(tau, q_perm=order, act_perm=act)
    ctrl.sync()
    assert (st.cpu().numpy() == 0).all()
    # oracle on the states the wire carried (float32), canonical order, torques re-indexed like the reference's S'u
    q32 = q.cpu().numpy(); v32 = v.cpu().numpy()
    qc = q32.copy(); vc = v32.copy()
    qc[7:] = q32[7 + np.array(order)]; vc[6:] = v32[6 + np.array(order)]
    tau_o, _, st_o = orc.step_batch("id", orc.model(b["model"]), orc.params("id"), qc, vc, b["targets"], b["mask"])
    raw = out.cpu().numpy().tobytes()
    for i in range(64):
        _, _, t = to.robot_state_decode(raw[204 * i:204 * i + 204])
        expect = np.zeros(12); expect[np.array(order)] = tau_o[:, i]                       # plant joint order
        assert np.abs(t - expect.astype(np.float32)).max() <= 2e-6 * max(np.abs(expect).max(), 1.0), i
    ctrl.close()
