"""tf.train.Saver V2 checkpoints without TensorFlow (gan_lib_tensorflow_amd/common/tf_checkpoint.py): container format,
checksums, name map, optimistic_restore.  No TensorFlow and no published checkpoint are available in this pipeline, so the
format is pinned by known answers that do not depend on this package's own writer: RFC 3720 CRC-32C vectors, LevelDB's
checksum mask, hand-assembled protobuf / block / footer bytes; the writer is then checked against the reader."""
import os
import struct

import numpy as np
import pytest

from gan_lib_tensorflow_amd.common import tf_checkpoint as C


def test_crc32c_known_answers_and_mask():
    assert C.crc32c(b"123456789") == 0xE3069283                      # the CRC catalogue's check value for CRC-32C
    assert C.crc32c(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4: 32 bytes of zeros
    assert C.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43                 # ... of ones
    assert C.crc32c(bytes(range(32))) == 0x46DD794E                   # ... incrementing
    rng = np.random.default_rng(0)
    for n in (65536, 70001, 300000):                                  # chunked numpy path == byte loop
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert C.crc32c(d) == C._crc32c_bytes(d)
    # leveldb/util/crc32c.h Mask(): rotate right by 15, add 0xa282ead8
    assert C._mask(0) == 0xa282ead8 and C._mask(0x00008000) == (1 + 0xa282ead8)


def test_entry_proto_and_block_layout_by_hand():
    # BundleEntryProto { dtype: DT_FLOAT  shape { dim {size: 2} dim {size: 3} }  size: 24  crc32c: 0x01020304 }
    want = bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03, 0x28, 0x18, 0x35, 0x04, 0x03, 0x02, 0x01])
    assert C._ser_entry(1, (2, 3), 0, 0, 24, 0x01020304) == want
    e = C._parse_entry(want + bytes([0x18, 0x02, 0x20, 0xAC, 0x02]))          # + shard_id: 2, offset: 300
    assert (e['dtype'], e['shape'], e['size'], e['crc32c'], e['shard_id'], e['offset']) == (1, [2, 3], 24, 0x01020304, 2, 300)
    # a LevelDB block: keys "ab", "abc" (shares 2 bytes), "b" with restart interval 2 -> restarts at entries 0 and 2
    blk = C._build_block([(b"ab", b"1"), (b"abc", b"22"), (b"b", b"")], restart_interval=2)
    assert blk == b"\x00\x02\x01ab1" + b"\x02\x01\x02c22" + b"\x00\x01\x00b" + struct.pack("<III", 0, 12, 2)


def test_write_read_round_trip_multi_block(tmp_path):
    rng = np.random.default_rng(1)
    tensors = {f"Generator/G.Block.{i}.Conv{j}/Filters": rng.normal(size=(3, 3, 8, 4 + i)).astype(np.float32) for i in range(200) for j in (1, 2)}
    tensors["Generator/G.Input/W"] = rng.normal(size=(128, 513)).astype(np.float32)       # > 64 KB: chunked checksum path
    tensors["beta2_power"] = np.asarray(0.9 ** 7, dtype=np.float32)
    tensors["global_step"] = np.asarray(12345, dtype=np.int64)
    tensors["labels"] = np.arange(10, dtype=np.int32)
    prefix = str(tmp_path / "model.ckpt-7")
    C.write_checkpoint(prefix, tensors)
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw) > 2 * 4096          # several data blocks (4 KB each)
    got = C.read_checkpoint(prefix)
    assert list(got) == sorted(tensors)                                                           # SSTable order
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    names = {n: (s, d) for n, s, d in C.list_variables(prefix)}
    assert names["Generator/G.Input/W"] == ((128, 513), np.dtype("<f4")) and names["global_step"] == ((), np.dtype("<i8"))
    only = C.read_checkpoint(prefix, names={"labels"})
    assert list(only) == ["labels"]
    # a flipped data byte is caught by the tensor checksum, a flipped index byte by the block checksum
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[100] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError, match="checksum"):
        C.read_checkpoint(prefix)
    assert C.read_checkpoint(prefix, verify=False)                                               # still readable when asked to
    idx = bytearray(raw)
    idx[50] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="checksum"):
        C.list_variables(prefix)


def test_name_map_and_optimistic_restore_into_a_store(tmp_path):
    from gan_lib_tensorflow_amd.store import ParamStore
    rng = np.random.default_rng(2)
    saved = {
        "Generator/G.Input/W": rng.normal(size=(4, 6)).astype(np.float32),
        "Generator/G.Input/W/Adam": rng.normal(size=(4, 6)).astype(np.float32),
        "Generator/G.Input/W/Adam_1": rng.random(size=(4, 6)).astype(np.float32),
        "Generator/G.Input/b": rng.normal(size=(6,)).astype(np.float32),
        "Discriminator/D.Output/W": rng.normal(size=(5, 1)).astype(np.float32),            # shape differs from the store's
        "Discriminator/D.Gone/W": rng.normal(size=(2, 2)).astype(np.float32),              # not in the store at all
        "beta1_power": np.asarray(0.0, np.float32), "beta2_power": np.asarray(0.9 ** 11, np.float32),
        "beta1_power_1": np.asarray(0.0, np.float32), "beta2_power_1": np.asarray(0.9 ** 55, np.float32),
    }
    prefix = str(tmp_path / "model.ckpt")
    C.write_checkpoint(prefix, saved)
    st = C.trainer_state_from_checkpoint(C.read_checkpoint(prefix))
    assert int(st["Generator/adam_t"]) == 11 and int(st["Discriminator/adam_t"]) == 55 and "beta2_power" not in st
    back = C.checkpoint_from_trainer_state(st)
    assert abs(float(back["beta2_power_1"]) - 0.9 ** 55) < 1e-9 and float(back["beta1_power"]) == 0.0
    store = ParamStore("cpu")
    w = store.get_variable("Generator/G.Input/W", None, np.zeros((4, 6), np.float32))
    b = store.get_variable("Generator/G.Input/b", None, np.zeros(6, np.float32))
    d = store.get_variable("Discriminator/D.Output/W", None, np.ones((128, 1), np.float32))
    restored = C.optimistic_restore(store, prefix)
    assert sorted(restored) == ["Generator/G.Input/W", "Generator/G.Input/b"]                    # name AND shape must match (misc.py:290-296)
    assert np.array_equal(w.detach().numpy(), saved["Generator/G.Input/W"]) and np.array_equal(b.detach().numpy(), saved["Generator/G.Input/b"])
    assert float(d.detach().sum()) == 128.0


def test_adam_step_count_survives_the_float32_underflow_of_beta2_power(tmp_path):
    """TF keeps beta2 ** t as float32: exactly 0.0 after ~980 steps (under 200 critic iterations).  A checkpoint of the
    reference then carries no count; the importer must treat it as SATURATED (bias correction = 1, what TF computes),
    never as a fresh optimiser, and our own writer/reader round trip must keep the exact count."""
    assert np.float32(0.9) ** np.float32(5000) == 0.0
    t_sat = C.adam_t_from_beta2_power(0.0)
    assert t_sat >= 900 and 1.0 - 0.9 ** t_sat == 1.0                       # sqrt(1 - beta2^t) is exactly 1
    assert C.adam_t_from_beta2_power(1.0) == 0                              # a fresh optimiser
    assert C.adam_t_from_beta2_power(np.float32(0.9 ** 55)) == 55
    assert C.adam_t_from_beta2_power(np.float32(1e-42)) == t_sat            # denormal: few bits left, count not recoverable
    # a reference-written checkpoint late in training: powers are 0.0f
    st = C.trainer_state_from_checkpoint({"beta2_power": np.asarray(0.0, np.float32), "beta2_power_1": np.asarray(0.0, np.float32)})
    assert int(st["Generator/adam_t"]) == t_sat and int(st["Discriminator/adam_t"]) == t_sat
    # our own state at t = 5000 / 25000: through write_checkpoint / read_checkpoint and back, exactly
    state = {"Generator/adam_t": np.asarray(5000, np.int64), "Discriminator/adam_t": np.asarray(25000, np.int64),
             "Generator/G.Input/b": np.zeros(3, np.float32)}
    tensors = C.checkpoint_from_trainer_state(state)
    assert float(tensors["beta2_power"]) == 0.0 and float(tensors["beta2_power_1"]) == 0.0      # what TF itself would hold
    prefix = str(tmp_path / "late.ckpt")
    C.write_checkpoint(prefix, tensors)
    back = C.trainer_state_from_checkpoint(C.read_checkpoint(prefix))
    assert int(back["Generator/adam_t"]) == 5000 and int(back["Discriminator/adam_t"]) == 25000
    assert back["Generator/adam_t"].dtype == np.int64
