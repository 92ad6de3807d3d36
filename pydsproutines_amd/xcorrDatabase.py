"""Results store of the correlator: the schema of the reference's ``xcorrDatabase.XcorrDB``
(xcorrDatabase/_core.py:24-118) on the standard library's sqlite3.

The reference builds its class on ``sew`` (the author's sqlite3 wrapper, an un-vendored dependency that is not present
here); this module keeps what a user of the class touches -- the table names, the column names / types / UNIQUE
constraints of the metadata table and of the three result-table types, ``createXcorrResultsTable``, ``reloadTables``,
``db[tblname]`` with ``fc / fs / s1 / s2 / xctype / desc``, ``insertOne / insertMany / select`` + ``db.fetchone()``,
and the ``regenerate*`` helpers (:239-266) -- so that files written by either side open with the other.

    type 0  one row of scalars per correlation: qf2, td, td_sigma, fd, fd_sigma, rfd, rfd_sigma
    type 1  per-delay results as blobs: qf2 (float64), freqIdx (uint32), rfdIdx
    type 2  the whole CAF as one blob

``store_peaks / store_rows / store_surface`` feed the three types straight from a device-resident ``CAFResult``:
the peak table, the per-delay (max, argmax) traces, the surface.  Host-side only; no GPU work happens here beyond the
device-to-host copy of what is stored.
"""

import sqlite3
from copy import deepcopy

import numpy as np


class XcorrDB:
    xcorr_metadata_tblname = "xcorr_metadata"
    xcorr_metadata_fmt = {
        "cols": [
            ["data_tblname", "TEXT"],
            ["fc", "REAL"],
            ["fs", "INTEGER"],
            ["s1", "TEXT"],
            ["s2", "TEXT"],
            ["xctype", "INTEGER"],
            ["desc", "BLOB"],
        ],
        "conds": ["UNIQUE(data_tblname)"],
        "foreign_keys": [],
    }
    xcorr_results_fmt = {
        "cols": [
            ["time_sec", "INTEGER"],
            ["tidx", "INTEGER"],
            ["cutoutlen", "INTEGER"],
            ["td_scan_start", "REAL"],
            ["td_scan_numsteps", "INTEGER"],
            ["td_scan_step", "REAL"],
            ["fd_scan_start", "REAL"],
            ["fd_scan_numsteps", "INTEGER"],
            ["fd_scan_step", "REAL"],
            ["rfd_scan_start", "REAL"],
            ["rfd_scan_numsteps", "INTEGER"],
            ["rfd_scan_step", "REAL"],
            ["desc", "BLOB"],
        ],
        "conds": [
            "UNIQUE(time_sec, tidx, cutoutlen, td_scan_start, td_scan_numsteps, td_scan_step, fd_scan_start, "
            "fd_scan_numsteps, fd_scan_step, rfd_scan_start, rfd_scan_numsteps, rfd_scan_step, desc)"
        ],
        "foreign_keys": [],
    }
    TYPE_PEAKVALUES = 0
    TYPE_1D = 1
    TYPE_2D = 2

    @staticmethod
    def _xcorr_type0results_fmt():
        fmt = deepcopy(XcorrDB.xcorr_results_fmt)
        fmt["cols"].extend([["qf2", "REAL"], ["td", "REAL"], ["td_sigma", "REAL"], ["fd", "REAL"], ["fd_sigma", "REAL"],
                            ["rfd", "REAL"], ["rfd_sigma", "REAL"]])
        return fmt

    @staticmethod
    def _xcorr_type1results_fmt():
        fmt = deepcopy(XcorrDB.xcorr_results_fmt)
        fmt["cols"].extend([["qf2", "BLOB"], ["freqIdx", "BLOB"], ["rfdIdx", "BLOB"]])
        return fmt

    @staticmethod
    def _xcorr_type2results_fmt():
        fmt = deepcopy(XcorrDB.xcorr_results_fmt)
        fmt["cols"].extend([["caf", "BLOB"]])
        return fmt

    # ------------------------------------------------------------------------------------------------------------
    def __init__(self, path=":memory:"):
        self.con = sqlite3.connect(path)
        self.con.row_factory = sqlite3.Row
        self.cur = self.con.cursor()
        self._tables = {}
        self._createMetaXcorrsTable()
        self.reloadTables()

    def close(self):
        self.con.close()

    def commit(self):
        self.con.commit()

    def fetchone(self):
        return self.cur.fetchone()

    def fetchall(self):
        return self.cur.fetchall()

    @staticmethod
    def _create_stmt(name, fmt, if_not_exists):
        cols = ", ".join('"%s" %s' % (c, t) for c, t in fmt["cols"])
        tail = "".join(", " + c for c in fmt["conds"] + fmt["foreign_keys"])
        return 'CREATE TABLE %s"%s" (%s%s)' % ("IF NOT EXISTS " if if_not_exists else "", name, cols, tail)

    def _createMetaXcorrsTable(self):
        self.cur.execute(self._create_stmt(self.xcorr_metadata_tblname, self.xcorr_metadata_fmt, True))
        self.con.commit()

    def reloadTables(self):
        self._tables = {}
        names = [r[0] for r in self.con.execute("SELECT name FROM sqlite_master WHERE type='table'")]
        meta = {}
        if self.xcorr_metadata_tblname in names:
            for r in self.con.execute('SELECT * FROM "%s"' % self.xcorr_metadata_tblname):
                meta[r["data_tblname"]] = r
        for n in names:
            if n == self.xcorr_metadata_tblname:
                self._tables[n] = XcorrMetaTableProxy(self, n)
            elif n in meta:
                self._tables[n] = XcorrResultsTableProxy(self, n)

    @property
    def tables(self):
        return self._tables

    def __getitem__(self, name):
        return self._tables[name]

    def createXcorrResultsTable(self, results_tblname, fc, fs, s1, s2, xctype, desc=None, ifNotExists=False):
        """ref: xcorrDatabase/_core.py:163-211: a results table of type 0 / 1 / 2 plus its row in the metadata table."""
        if xctype == 0:
            fmt = XcorrDB._xcorr_type0results_fmt()
        elif xctype == 1:
            fmt = XcorrDB._xcorr_type1results_fmt()
        elif xctype == 2:
            fmt = XcorrDB._xcorr_type2results_fmt()
        else:
            raise ValueError(f"xctype must be 0, 1, or 2, not {xctype}")
        self.cur.execute(self._create_stmt(results_tblname, fmt, ifNotExists))
        self.cur.execute('INSERT OR REPLACE INTO "%s" VALUES (?,?,?,?,?,?,?)' % self.xcorr_metadata_tblname,
                         (results_tblname, float(fc), int(fs), s1, s2, int(xctype), desc))
        self.con.commit()
        self.reloadTables()
        return self._tables[results_tblname]

    # ---- feeders from the engine's outputs ----------------------------------------------------------------------
    def store_peaks(self, tblname, result, freqs_hz, fs, time_sec=0, tidx=0, cutoutlen=0, shift_start=0, desc=None):
        """Type 0: one row per template of a CAFResult's peak table (peak delay -> td in seconds, peak frequency index
        -> fd in Hz through ``freqs_hz``)."""
        tbl = self._tables[tblname]
        if tbl.xctype != self.TYPE_PEAKVALUES:
            raise ValueError("table %s is not of type 0" % tblname)
        pv, pd, pf = result.peak_val.get(), result.peak_delay.get(), result.peak_freq.get()
        fr = np.asarray(freqs_hz, dtype=np.float64)
        rows = [{"time_sec": int(time_sec), "tidx": int(tidx) + t, "cutoutlen": int(cutoutlen), "td_scan_start": shift_start / fs,
                 "td_scan_step": 1.0 / fs, "fd_scan_start": float(fr[0]), "fd_scan_numsteps": int(fr.size),
                 "fd_scan_step": float(fr[1] - fr[0]) if fr.size > 1 else 0.0, "desc": desc, "qf2": float(pv[t]),
                 "td": float(pd[t]) / fs, "fd": float(fr[pf[t]])} for t in range(pv.size)]
        tbl.insertMany(rows, commitNow=True)
        return len(rows)

    def store_rows(self, tblname, result, fs, template=0, time_sec=0, tidx=0, cutoutlen=0, shift_start=0, fd_scan=(0.0, 0, 0.0),
                   desc=None):
        """Type 1: the per-delay traces of one template (qf2 as float64, freqIdx as uint32: the dtypes the schema
        recommends, :97-99, and fastXcorr returns)."""
        tbl = self._tables[tblname]
        if tbl.xctype != self.TYPE_1D:
            raise ValueError("table %s is not of type 1" % tblname)
        q = result.row_max[template].get().astype(np.float64)
        f = result.row_arg[template].get().astype(np.uint32)
        tbl.insertOne({"time_sec": int(time_sec), "tidx": int(tidx), "cutoutlen": int(cutoutlen), "td_scan_start": shift_start / fs,
                       "td_scan_numsteps": int(q.size), "td_scan_step": 1.0 / fs, "fd_scan_start": float(fd_scan[0]),
                       "fd_scan_numsteps": int(fd_scan[1]), "fd_scan_step": float(fd_scan[2]), "desc": desc,
                       "qf2": q.tobytes(), "freqIdx": f.tobytes()}, commitNow=True)

    def store_surface(self, tblname, result, fs, template=0, time_sec=0, tidx=0, cutoutlen=0, shift_start=0, fd_scan=(0.0, 0, 0.0),
                      desc=None):
        """Type 2: the CAF surface of one template as a float32 blob of shape (td_scan_numsteps, fd_scan_numsteps)."""
        tbl = self._tables[tblname]
        if tbl.xctype != self.TYPE_2D:
            raise ValueError("table %s is not of type 2" % tblname)
        s = result.surface[template].get()
        tbl.insertOne({"time_sec": int(time_sec), "tidx": int(tidx), "cutoutlen": int(cutoutlen), "td_scan_start": shift_start / fs,
                       "td_scan_numsteps": int(s.shape[0]), "td_scan_step": 1.0 / fs, "fd_scan_start": float(fd_scan[0]),
                       "fd_scan_numsteps": int(s.shape[1]), "fd_scan_step": float(fd_scan[2]), "desc": desc, "caf": s.tobytes()},
                      commitNow=True)


class _TableProxy:
    def __init__(self, parent, name):
        self._parent, self._tbl = parent, name
        self._cols = [r["name"] for r in parent.con.execute('PRAGMA table_info("%s")' % name)]

    @property
    def columnNames(self):
        return list(self._cols)

    def select(self, columnNames="*", conditions=None, orderBy=None):
        """Runs the SELECT on the parent's cursor; read with ``db.fetchone()`` / ``db.fetchall()`` (sew's convention)."""
        cols = columnNames if isinstance(columnNames, str) else ", ".join('"%s"' % c for c in columnNames)
        stmt = 'SELECT %s FROM "%s"' % (cols, self._tbl)
        if conditions:
            stmt += " WHERE " + (conditions if isinstance(conditions, str) else " AND ".join(conditions))
        if orderBy:
            stmt += " ORDER BY " + (orderBy if isinstance(orderBy, str) else ", ".join(orderBy))
        self._parent.cur.execute(stmt)
        return stmt

    def insertOne(self, row, orReplace=False, commitNow=False):
        """``row``: dict column -> value (missing columns are NULL), or a full sequence of values."""
        if isinstance(row, dict):
            unknown = [k for k in row if k not in self._cols]
            if unknown:
                raise KeyError("unknown columns %r" % unknown)
            keys = list(row)
            stmt = 'INSERT%s INTO "%s" (%s) VALUES (%s)' % (" OR REPLACE" if orReplace else "", self._tbl,
                                                         ", ".join('"%s"' % k for k in keys), ",".join("?" * len(keys)))
            self._parent.cur.execute(stmt, [row[k] for k in keys])
        else:
            stmt = 'INSERT%s INTO "%s" VALUES (%s)' % (" OR REPLACE" if orReplace else "", self._tbl, ",".join("?" * len(self._cols)))
            self._parent.cur.execute(stmt, list(row))
        if commitNow:
            self._parent.con.commit()

    def insertMany(self, rows, orReplace=False, commitNow=False):
        for r in rows:
            self.insertOne(r, orReplace=orReplace)
        if commitNow:
            self._parent.con.commit()


class XcorrMetaTableProxy(_TableProxy):
    def getMetadataFor(self, data_tblname):
        r = self._parent.con.execute('SELECT * FROM "%s" WHERE data_tblname = ?' % self._tbl, (data_tblname,)).fetchone()
        if r is None:
            raise KeyError(data_tblname)
        return r


class XcorrResultsTableProxy(_TableProxy):
    """ref: xcorrDatabase/_core.py:218-266."""

    def __init__(self, parent, name):
        super().__init__(parent, name)
        self._cacheMetadata = None

    def getMetadata(self):
        return self._parent[XcorrDB.xcorr_metadata_tblname].getMetadataFor(self._tbl)

    def _meta(self, key):
        self._cacheMetadata = self.getMetadata() if self._cacheMetadata is None else self._cacheMetadata
        return self._cacheMetadata[key]

    fc = property(lambda self: self._meta("fc"))
    fs = property(lambda self: self._meta("fs"))
    s1 = property(lambda self: self._meta("s1"))
    s2 = property(lambda self: self._meta("s2"))
    xctype = property(lambda self: self._meta("xctype"))
    desc = property(lambda self: self._meta("desc"))

    def regenerateTDscanRange(self, td_scan_start, td_scan_numsteps, td_scan_step):
        return np.arange(td_scan_numsteps) * td_scan_step + td_scan_start

    def regenerate1Darray(self, qf2, dtype=np.float64):
        return np.frombuffer(qf2, dtype=dtype)

    def regenerate1Dresults(self, row, qf2type=np.float64, freqindtype=np.uint32):
        tdrange = self.regenerateTDscanRange(row["td_scan_start"], row["td_scan_numsteps"], row["td_scan_step"])
        return tdrange, self.regenerate1Darray(row["qf2"], dtype=qf2type), self.regenerate1Darray(row["freqIdx"], dtype=freqindtype)

    def regenerate2Dresults(self, row, dtype=np.float32):
        return np.frombuffer(row["caf"], dtype=dtype).reshape(row["td_scan_numsteps"], row["fd_scan_numsteps"])


__all__ = ["XcorrDB", "XcorrMetaTableProxy", "XcorrResultsTableProxy"]
