"""The results store (pydsproutines_amd/xcorrDatabase.py) against the reference's own unit test
(xcorrDatabase/_core.py:271-377, restated: metadata table present, a type-0 table with its metadata, one inserted
row read back) and its schema text (:28-117); plus the blob round trips of types 1 and 2.  CPU only; the feeders from
a device-resident CAFResult are covered by the GPU test at the end."""

import numpy as np
import pytest

from pydsproutines_amd.xcorrDatabase import XcorrDB


def test_reference_unit_test_restated():
    db = XcorrDB(":memory:")
    assert db.xcorr_metadata_tblname in db.tables
    db.createXcorrResultsTable("results", 1e9, 1000, "source1", "source2", 0)
    db.reloadTables()
    assert "results" in db.tables
    results = db["results"]
    assert (results.fc, results.fs, results.s1, results.s2, results.xctype, results.desc) == (1e9, 1000, "source1", "source2", 0, None)
    inserted = {"time_sec": 1234567890, "tidx": 123, "cutoutlen": 10000, "td_scan_start": -1e-3, "td_scan_numsteps": 10000,
                "td_scan_step": 1e-6, "qf2": 0.9, "td": 0, "td_sigma": 1e-7}
    results.insertOne(inserted, commitNow=True)
    results.select("*")
    r = db.fetchone()
    for k in inserted:
        assert r[k] == inserted[k]
    with pytest.raises(ValueError):
        db.createXcorrResultsTable("bad", 1e9, 1000, "a", "b", 3)


def test_schema_columns_match_the_reference_text():
    base = ["time_sec", "tidx", "cutoutlen", "td_scan_start", "td_scan_numsteps", "td_scan_step", "fd_scan_start", "fd_scan_numsteps",
            "fd_scan_step", "rfd_scan_start", "rfd_scan_numsteps", "rfd_scan_step", "desc"]
    db = XcorrDB()
    t0 = db.createXcorrResultsTable("t0", 0.0, 1, "a", "b", XcorrDB.TYPE_PEAKVALUES)
    t1 = db.createXcorrResultsTable("t1", 0.0, 1, "a", "b", XcorrDB.TYPE_1D)
    t2 = db.createXcorrResultsTable("t2", 0.0, 1, "a", "b", XcorrDB.TYPE_2D)
    assert t0.columnNames == base + ["qf2", "td", "td_sigma", "fd", "fd_sigma", "rfd", "rfd_sigma"]
    assert t1.columnNames == base + ["qf2", "freqIdx", "rfdIdx"]
    assert t2.columnNames == base + ["caf"]
    assert db[db.xcorr_metadata_tblname].columnNames == ["data_tblname", "fc", "fs", "s1", "s2", "xctype", "desc"]
    # the UNIQUE condition over the scan description is in force
    row = {"time_sec": 1, "tidx": 2, "cutoutlen": 3, "td_scan_start": 0.0, "td_scan_numsteps": 4, "td_scan_step": 1.0,
           "fd_scan_start": 0.0, "fd_scan_numsteps": 0, "fd_scan_step": 0.0, "rfd_scan_start": 0.0, "rfd_scan_numsteps": 0,
           "rfd_scan_step": 0.0, "desc": b"x", "qf2": 0.5}
    t0.insertOne(row, commitNow=True)
    import sqlite3
    with pytest.raises(sqlite3.IntegrityError):
        t0.insertOne(row, commitNow=True)


def test_blob_round_trips(tmp_path):
    path = str(tmp_path / "x.db")
    db = XcorrDB(path)
    t1 = db.createXcorrResultsTable("rows", 2.4e9, 4096, "tx", "rx", 1)
    q = np.linspace(0, 1, 50)
    f = np.arange(50, dtype=np.uint32)
    t1.insertOne({"time_sec": 7, "td_scan_start": 1.5, "td_scan_numsteps": 50, "td_scan_step": 0.25, "qf2": q.tobytes(),
                  "freqIdx": f.tobytes()}, commitNow=True)
    t2 = db.createXcorrResultsTable("caf", 2.4e9, 4096, "tx", "rx", 2)
    s = np.arange(12, dtype=np.float32).reshape(4, 3)
    t2.insertOne({"time_sec": 7, "td_scan_numsteps": 4, "fd_scan_numsteps": 3, "caf": s.tobytes()}, commitNow=True)
    db.close()
    db = XcorrDB(path)  # reopen: the tables are rediscovered through the metadata table
    assert db["rows"].xctype == 1 and db["caf"].fs == 4096
    db["rows"].select("*")
    td, q2, f2 = db["rows"].regenerate1Dresults(db.fetchone())
    np.testing.assert_array_equal(q2, q)
    np.testing.assert_array_equal(f2, f)
    np.testing.assert_allclose(td, 1.5 + 0.25 * np.arange(50))
    db["caf"].select("*")
    np.testing.assert_array_equal(db["caf"].regenerate2Dresults(db.fetchone()), s)


@pytest.mark.gpu
def test_feeders_from_a_device_result():
    from conftest import cn, qpsk
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(9)
    n, m, fs = 256, 30_000, 256.0
    tm = np.stack([qpsk(rng, n) for _ in range(3)])
    rx = cn(rng, m)
    truth = [(4000, 5), (12_345, -3), (25_000, 0)]
    for t, (d, k) in enumerate(truth):
        rx[d : d + n] += (tm[t] * np.exp(2j * np.pi * k * np.arange(n) / n)).astype(np.complex64)
    bins = np.arange(-8, 8)
    res = CAFPlan(tm, max_rx_len=m, bins=bins, grid=n).run(asarray(rx), surface=True, rows=True, peak=True)
    db = XcorrDB()
    db.createXcorrResultsTable("peaks", 0.0, int(fs), "templates", "rx", 0)
    db.createXcorrResultsTable("rows", 0.0, int(fs), "templates", "rx", 1)
    db.createXcorrResultsTable("caf", 0.0, int(fs), "templates", "rx", 2)
    assert db.store_peaks("peaks", res, bins * fs / n, fs, cutoutlen=n) == 3
    db.store_rows("rows", res, fs, template=1, cutoutlen=n, fd_scan=(float(bins[0]), bins.size, 1.0))
    db.store_surface("caf", res, fs, template=2, cutoutlen=n, fd_scan=(float(bins[0]), bins.size, 1.0))
    db["peaks"].select("*", orderBy="tidx")
    for (d, k), r in zip(truth, db.fetchall()):
        assert round(r["td"] * fs) == d and r["fd"] == k and 0.3 < r["qf2"] < 0.7
    db["rows"].select("*")
    _, q, f = db["rows"].regenerate1Dresults(db.fetchone())
    np.testing.assert_array_equal(q, res.row_max[1].get().astype(np.float64))
    assert int(np.argmax(q)) == truth[1][0] and bins[f[truth[1][0]]] == truth[1][1]
    db["caf"].select("*")
    np.testing.assert_array_equal(db["caf"].regenerate2Dresults(db.fetchone()), res.surface[2].get())
