"""Host-side mirror of the reference's R entry points for the PeakSegFPOP path.

R is not installed in this image, so the layer the reference keeps in R
(/root/reference/R/*.R) is mirrored here in Python with the same names, argument meaning,
file protocol and error messages; the native call goes through the same C ABI the R glue
would use.  Data frames are pandas.DataFrame, R lists are plain classes.

  PeakSegFPOP_file     <- R/PeakSegFPOP_file.R:30-87
  PeakSegFPOP_dir      <- R/PeakSegFPOP_dir.R:47-117   (+ coef / summary, :215-238)
  PeakSegFPOP_df       <- R/PeakSegFPOP_df.R:18-35
  PeakSegFPOP_vec      <- R/PeakSegFPOP_vec.R:9-25
  sequentialSearch_dir <- R/sequentialSearch_dir.R:22-103
  writeBedGraph        <- R/writeBedGraph.R:8-38
  col_name_list        <- R/col.name.list.R:10-18
"""
import math
import os
import shutil
import tempfile
import time

import numpy as np
import pandas as pd

from . import _native

col_name_list = {
    "loss": ["penalty", "segments", "peaks", "bases", "bedGraph.lines", "mean.pen.cost",
             "total.loss", "equality.constraints", "mean.intervals", "max.intervals"],
    "segments": ["chrom", "chromStart", "chromEnd", "status", "mean"],
    "coverage": ["chrom", "chromStart", "chromEnd", "count"],
}


class PeakSegError(RuntimeError):
    """Raised where the reference's glue calls Rf_error (src/interface.cpp:16-55)."""

    def __init__(self, status, message):
        RuntimeError.__init__(self, message)
        self.status = status


def paste(x):
    """R's paste()/as.character() of a scalar: doubles use up to 15 significant digits and
    the narrower of fixed / scientific notation (R formatReal with digits=15)."""
    if isinstance(x, str):
        return x
    if isinstance(x, (bool, np.bool_)):
        return "TRUE" if x else "FALSE"
    if isinstance(x, (int, np.integer)):
        return "%d" % x
    x = float(x)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    if x == 0:
        return "0"
    # minimal number of significant digits (<= 15) that reproduces the 15-digit value
    target = float("%.14e" % x)
    nsig = 15
    for d in range(1, 16):
        if float("%.*e" % (d - 1, x)) == target:
            nsig = d
            break
    mant, exp = ("%.*e" % (nsig - 1, x)).split("e")
    kpower = int(exp)
    neg = 1 if x < 0 else 0
    if kpower >= 0:
        left = kpower + 1
        rgt = max(0, nsig - kpower - 1)
    else:
        left = 1
        rgt = nsig - kpower - 1
    w_fixed = neg + left + (rgt + 1 if rgt > 0 else 0)
    w_sci = neg + (nsig + 1 if nsig > 1 else 1) + (5 if abs(kpower) >= 100 else 4)
    if w_fixed <= w_sci:
        return "%.*f" % (rgt, x)
    return "%.*e" % (nsig - 1, x)


# ---- writeBedGraph (R/writeBedGraph.R:8-38) ------------------------------------------------

def writeBedGraph(count_df, coverage_bedGraph):
    if not isinstance(count_df, pd.DataFrame):
        raise ValueError("count.df must be data.frame")
    exp_names = ["chrom", "chromStart", "chromEnd", "count"]
    if list(count_df.columns) != exp_names:
        raise ValueError("count.df must have names " + ", ".join(exp_names))
    if not pd.api.types.is_integer_dtype(count_df["chromStart"]):
        raise ValueError("count.df$chromStart must be integer")
    if not pd.api.types.is_integer_dtype(count_df["chromEnd"]):
        raise ValueError("count.df$chromEnd must be integer")
    if not pd.api.types.is_numeric_dtype(count_df["count"]):
        raise ValueError("count.df$count must be numeric")
    if (count_df["chromStart"] < 0).any():
        raise ValueError("count.df$chromStart must always be non-negative")
    if not (count_df["chromStart"] < count_df["chromEnd"]).all():
        raise ValueError("chromStart must be less than chromEnd for all rows of count.df")
    with open(coverage_bedGraph, "w") as f:
        for chrom, s, e, c in zip(count_df["chrom"], count_df["chromStart"],
                                  count_df["chromEnd"], count_df["count"]):
            f.write("%s\t%d\t%d\t%s\n" % (chrom, s, e, paste(c)))


# ---- PeakSegFPOP_file (R/PeakSegFPOP_file.R:30-87) -----------------------------------------

def _native_interface(bedGraph_file, pen_str, db_file):
    """`.C("PeakSegFPOP_interface", ...)`: run the solver, turn a status into the
    reference's error text (src/interface.cpp:10-56)."""
    status = _native.lib.PeakSegFPOP_disk(
        os.fsencode(bedGraph_file), pen_str.encode(), os.fsencode(db_file))
    if status != 0:
        msg = _native.status_message(status, bedGraph_file, pen_str, db_file)
        detail = _native.last_error()
        if status >= _native.ERROR_NO_HIP_DEVICE and detail:
            msg = "%s (%s)" % (msg, detail)
        raise PeakSegError(status, msg)


def PeakSegFPOP_file(bedGraph_file, pen_str, db_file=None):
    if not (isinstance(bedGraph_file, str) and os.path.exists(bedGraph_file)):
        raise ValueError("bedGraph.file=%s must be the name of a data file to segment"
                         % (bedGraph_file,))
    if not isinstance(pen_str, str):
        raise ValueError("pen.str must be a character string that can be converted to a "
                         "non-negative numeric scalar")
    try:
        penalty = float(pen_str)  # as.numeric(pen.str); "Inf" is fine
    except ValueError:
        penalty = float("nan")
    if not (0 <= penalty <= float("inf")):
        raise ValueError("as.numeric(pen.str)=%s but it must be a non-negative numeric scalar"
                         % paste(penalty))
    norm_file = os.path.realpath(bedGraph_file)
    if db_file is None:
        db_file = "%s_penalty=%s.db" % (norm_file, pen_str)
    if not isinstance(db_file, str):
        raise ValueError("db.file=%s must be a temporary file name where cost function db "
                         "can be written" % (db_file,))
    if os.path.isfile(db_file):
        os.unlink(db_file)
    _native_interface(norm_file, pen_str, db_file)
    result = {"bedGraph.file": norm_file, "penalty": pen_str, "db.file": db_file}
    result["megabytes"] = (os.path.getsize(db_file) / 1024 / 1024
                           if os.path.isfile(db_file) else 0)
    if os.path.isfile(db_file):
        os.unlink(db_file)
    loss_tsv = "%s_penalty=%s_loss.tsv" % (bedGraph_file, pen_str)
    if os.path.getsize(loss_tsv) == 0:
        raise PeakSegError(8, "unable to write to loss output file %s (disk is probably full)"
                           % loss_tsv)
    return result


# ---- PeakSegFPOP_dir (R/PeakSegFPOP_dir.R:47-117) ------------------------------------------

class PeakSegFPOP_dir_result:
    """R list of class c("PeakSegFPOP_dir","list"): $segments, $loss (+ $data, $others)."""

    def __init__(self, segments, loss):
        self.segments = segments
        self.loss = loss
        self.data = None
        self.others = None
        self.classes = ["PeakSegFPOP_dir", "list"]

    def summary(self):  # summary.PeakSegFPOP_dir (R/PeakSegFPOP_dir.R:234-238)
        return self.loss

    def coef(self):  # coef.PeakSegFPOP_dir (R/PeakSegFPOP_dir.R:215-231)
        seg = self.segments
        d = np.diff(seg["mean"].to_numpy())
        changes = pd.DataFrame({
            "type": "segmentation",
            "constraint": np.where(d == 0, "equality", "inequality"),
            "chromEnd": seg["chromEnd"].to_numpy()[1:]})
        peaks = seg[seg["status"] == "peak"].copy()
        peaks.insert(0, "type", "peaks")
        out = PeakSegFPOP_dir_result(seg.assign(type="segmentation")[["type"] + list(seg.columns)],
                                     self.loss)
        out.changes = changes
        out.peaks = peaks
        out.data = self.data
        return out


def _read_table(path, names):
    """fread(file=..., col.names=names) of one of the solver's tab-separated files."""
    if os.path.getsize(path) == 0:
        raise ValueError("empty file %s" % path)
    # (round_trip: the default parser of read_csv is off by an ulp on some 20-digit fields,
    # and a penalty read back from _loss.tsv must print as the string that named the file)
    return pd.read_csv(path, sep="\t", header=None, names=names, na_filter=False,
                       float_precision="round_trip")


def _first_last_line(path, names):
    with open(path, "rb") as f:
        first = f.readline()
        f.seek(0, os.SEEK_END)
        size = f.tell()
        if size == 0 or not first.strip():
            raise ValueError("empty file")
        back = min(size, 4096)
        f.seek(size - back)
        tail = f.read().splitlines()
        last = tail[-1] if tail[-1].strip() else tail[-2]

    def parse(line):
        vals = line.decode().split()
        if len(vals) != len(names):
            raise ValueError("bad column count")
        return dict(zip(names, vals))
    return parse(first), parse(last)


def _already_computed(prob_cov_bedGraph, segments_bed, loss_tsv, timing_tsv):
    """The cache predicate of R/PeakSegFPOP_dir.R:70-93 (any error means recompute)."""
    try:
        timing = _read_table(timing_tsv, ["penalty", "megabytes", "seconds"])
        first_seg, last_seg = _first_last_line(segments_bed, col_name_list["segments"])
        first_cov, last_cov = _first_last_line(prob_cov_bedGraph, col_name_list["coverage"])
        loss = _read_table(loss_tsv, col_name_list["loss"])
        nrow_ok = len(timing) == 1 and len(loss) == 1
        consistent = (int(first_seg["chromEnd"]) - int(last_seg["chromStart"])
                      == int(loss["bases"].iloc[0]))
        start_ok = int(first_cov["chromStart"]) == int(last_seg["chromStart"])
        end_ok = int(last_cov["chromEnd"]) == int(first_seg["chromEnd"])
        if nrow_ok and consistent and start_ok and end_ok:
            return timing, loss
    except Exception:
        pass
    return None


def PeakSegFPOP_dir(problem_dir, penalty_param, db_file=None):
    if not (isinstance(problem_dir, str) and os.path.isdir(problem_dir)):
        raise ValueError("problem.dir=%s must be the name of a directory containing a file "
                         "named coverage.bedGraph" % (problem_dir,))
    ok_type = isinstance(penalty_param, (int, float, str, np.integer, np.floating)) and \
        not isinstance(penalty_param, bool)
    if not ok_type or (isinstance(penalty_param, float) and math.isnan(penalty_param)):
        raise ValueError("penalty.param must be numeric or character, length 1, not missing")
    penalty_str = paste(penalty_param)
    prob_cov_bedGraph = os.path.join(problem_dir, "coverage.bedGraph")
    pre = "%s_penalty=%s" % (prob_cov_bedGraph, penalty_str)
    penalty_segments_bed = pre + "_segments.bed"
    penalty_loss_tsv = pre + "_loss.tsv"
    penalty_timing_tsv = pre + "_timing.tsv"
    cached = _already_computed(prob_cov_bedGraph, penalty_segments_bed, penalty_loss_tsv,
                               penalty_timing_tsv)
    if cached is None:
        t0 = time.time()
        result = PeakSegFPOP_file(prob_cov_bedGraph, penalty_str, db_file)
        seconds = time.time() - t0
        timing = pd.DataFrame({"penalty": [float(penalty_str)],
                               "megabytes": [result["megabytes"]], "seconds": [seconds]})
        with open(penalty_timing_tsv, "w") as f:
            f.write("%s\t%s\t%s\n" % (paste(float(penalty_str)), paste(result["megabytes"]),
                                      paste(seconds)))
        penalty_loss = _read_table(penalty_loss_tsv, col_name_list["loss"])
    else:
        timing, penalty_loss = cached
    penalty_segs = _read_table(penalty_segments_bed, col_name_list["segments"])
    loss = penalty_loss.copy()
    loss["megabytes"] = float(timing["megabytes"].iloc[0])
    loss["seconds"] = float(timing["seconds"].iloc[0])
    return PeakSegFPOP_dir_result(penalty_segs, loss)


# ---- PeakSegFPOP_df / _vec (R/PeakSegFPOP_df.R:18-35, R/PeakSegFPOP_vec.R:9-25) -------------

def _check_pen_num(pen_num):
    if not (isinstance(pen_num, (int, float, np.integer, np.floating))
            and not isinstance(pen_num, bool) and 0 <= pen_num):
        raise ValueError("pen.num must be non-negative numeric scalar")


def PeakSegFPOP_df(count_df, pen_num, base_dir=None):
    _check_pen_num(pen_num)
    if base_dir is None:
        base_dir = tempfile.gettempdir()
    data_dir = os.path.join(base_dir, "%s-%d-%d" % (
        count_df["chrom"].iloc[0], count_df["chromStart"].min(), count_df["chromEnd"].max()))
    shutil.rmtree(data_dir, ignore_errors=True)
    os.makedirs(data_dir, exist_ok=True)
    coverage_bedGraph = os.path.join(data_dir, "coverage.bedGraph")
    writeBedGraph(count_df, coverage_bedGraph)
    L = PeakSegFPOP_dir(data_dir, paste(pen_num))
    L.data = count_df.copy()
    L.classes = ["PeakSegFPOP_df"] + L.classes
    return L


def PeakSegFPOP_vec(count_vec, pen_num):
    _check_pen_num(pen_num)
    count_vec = np.asarray(count_vec)
    if not np.issubdtype(count_vec.dtype, np.integer):
        raise ValueError("count.vec must be integer")
    # rle(count.vec)
    change = np.flatnonzero(np.diff(count_vec) != 0)
    ends = np.concatenate([change + 1, [len(count_vec)]]).astype(np.int64)
    starts = np.concatenate([[0], ends[:-1]]).astype(np.int64)
    coverage_df = pd.DataFrame({"chrom": "chrUnknown", "chromStart": starts, "chromEnd": ends,
                                "count": count_vec[starts]})
    return PeakSegFPOP_df(coverage_df, pen_num)


# ---- PeakSegFPOP_dir for a batch (additive) --------------------------

def PeakSegFPOP_dir_batch(problem_dirs, penalty_params):
    """PeakSegFPOP_dir for many (problem.dir, penalty) pairs in one call of the native
    PeakSegFPOP_dir_batch: cached results are reused as PeakSegFPOP_dir would
    (R/PeakSegFPOP_dir.R:70-93), every other pair is solved in one device problem set (one
    parse and upload per distinct coverage.bedGraph) and gets its _timing.tsv.  Returns the
    list of PeakSegFPOP_dir results, in order; `.cached` says which were reused."""
    import ctypes
    if len(problem_dirs) != len(penalty_params):
        raise ValueError("problem_dirs and penalty_params must have the same length")
    n = len(problem_dirs)
    pens = [paste(p) for p in penalty_params]
    dirs = (ctypes.c_char_p * n)(*[os.fsencode(d) for d in problem_dirs])
    pstr = (ctypes.c_char_p * n)(*[p.encode() for p in pens])
    status = (ctypes.c_int * n)()
    cached = (ctypes.c_int * n)()
    _native.lib.PeakSegFPOP_dir_batch(n, dirs, pstr, status, cached)
    out = []
    for i in range(n):
        bg = os.path.join(problem_dirs[i], "coverage.bedGraph")
        if status[i] != 0:
            msg = _native.status_message(status[i], os.path.realpath(bg), pens[i],
                                         "%s_penalty=%s.db" % (os.path.realpath(bg), pens[i]))
            raise PeakSegError(status[i], msg)
        L = PeakSegFPOP_dir(problem_dirs[i], pens[i])  # reads the files just written (cache hit)
        L.cached = bool(cached[i])
        out.append(L)
    return out


# ---- sequentialSearch_dir (R/sequentialSearch_dir.R:22-103) --------------------------------

def sequentialSearch_dir(problem_dir, peaks_int, verbose=0):
    """The reference's penalty search for a target number of peaks.  The loop itself runs in
    the native library (PeakSegFPOP_sequential_search: coverage.bedGraph parsed and uploaded
    once, arena reused from one penalty to the next); it visits the reference's penalties and
    leaves the reference's files, from which the result is assembled here."""
    import ctypes
    if not (isinstance(peaks_int, (int, np.integer)) and not isinstance(peaks_int, bool)
            and 0 <= peaks_int):
        raise ValueError("is.integer(peaks.int) && length(peaks.int) == 1 && 0 <= peaks.int "
                         "is not TRUE")
    if not isinstance(problem_dir, str):
        raise ValueError("is.character(problem.dir) is not TRUE")
    cap = 256
    rows = (_native.PsdSearchRow * cap)()
    n_rows = ctypes.c_int(0)
    chosen = ctypes.c_int(-1)
    st = _native.lib.PeakSegFPOP_sequential_search(
        os.fsencode(problem_dir), int(peaks_int), int(bool(verbose)), cap, rows,
        ctypes.byref(n_rows), ctypes.byref(chosen))
    if st == _native.ERROR_SEARCH_TOO_MANY_PEAKS:
        raise ValueError(_native.last_error())
    if st != 0:
        bg = os.path.realpath(os.path.join(problem_dir, "coverage.bedGraph"))
        pen = rows[n_rows.value].penalty_str.decode() if n_rows.value < cap else ""
        msg = _native.status_message(st, bg, pen, "%s_penalty=%s.db" % (bg, pen))
        detail = _native.last_error()
        if st >= _native.ERROR_NO_HIP_DEVICE and detail:
            msg = "%s (%s)" % (msg, detail)
        raise PeakSegError(st, msg)
    return _search_result(problem_dir, [rows[k] for k in range(n_rows.value)], chosen.value)


def _search_result(problem_dir, rows, chosen):
    """sequentialSearch_dir's value from the rows of a native search: the chosen model with
    $others (R/sequentialSearch_dir.R:96-102)."""
    model_list = {}
    for r in rows:
        pen_str = r.penalty_str.decode()
        L = PeakSegFPOP_dir(problem_dir, pen_str)  # the files of this model (cache hit)
        L.loss["iteration"] = r.iteration
        L.loss["under"] = np.nan if r.under_peaks == _native.SEARCH_NA else r.under_peaks
        L.loss["over"] = np.nan if r.over_peaks == _native.SEARCH_NA else r.over_peaks
        model_list[pen_str] = L
    out = model_list[rows[chosen].penalty_str.decode()]
    others = pd.concat([m.loss for m in model_list.values()], ignore_index=True)
    out.others = others.sort_values("iteration", kind="stable").reset_index(drop=True)
    return out


def sequentialSearch_dir_batch(problem_dirs, peaks_int, verbose=0):
    """sequentialSearch_dir over several problem directories at once (additive): each directory
    gets the result sequentialSearch_dir(dir, peaks) gives, but the models the searches ask for
    in the same iteration are computed in one device launch
    (PeakSegFPOP_sequential_search_batch).  peaks_int: one target for all, or one per
    directory.  Returns the list of results; a failed search raises for the first failure."""
    import ctypes
    problem_dirs = list(problem_dirs)
    n = len(problem_dirs)
    if isinstance(peaks_int, (int, np.integer)) and not isinstance(peaks_int, bool):
        peaks_int = [int(peaks_int)] * n
    peaks_int = [int(p) for p in peaks_int]
    if len(peaks_int) != n or any(p < 0 for p in peaks_int):
        raise ValueError("peaks.int: one non-negative integer, or one per problem directory")
    if not all(isinstance(d, str) for d in problem_dirs):
        raise ValueError("is.character(problem.dir) is not TRUE")
    if n == 0:
        return []
    cap = 256
    rows = (_native.PsdSearchRow * (cap * n))()
    dirs = (ctypes.c_char_p * n)(*[os.fsencode(d) for d in problem_dirs])
    peaks = (ctypes.c_int * n)(*peaks_int)
    n_rows = (ctypes.c_int * n)()
    chosen = (ctypes.c_int * n)()
    status = (ctypes.c_int * n)()
    _native.lib.PeakSegFPOP_sequential_search_batch(n, dirs, peaks, int(bool(verbose)), cap, rows,
                                                    n_rows, chosen, status)
    out = []
    for d in range(n):
        if status[d] == _native.ERROR_SEARCH_TOO_MANY_PEAKS:
            raise ValueError(_native.last_error())
        if status[d] != 0:
            bg = os.path.realpath(os.path.join(problem_dirs[d], "coverage.bedGraph"))
            k = n_rows[d]
            pen = rows[d * cap + k].penalty_str.decode() if k < cap else ""
            raise PeakSegError(status[d], _native.status_message(
                status[d], bg, pen, "%s_penalty=%s.db" % (bg, pen)))
        out.append(_search_result(problem_dirs[d],
                                  [rows[d * cap + k] for k in range(n_rows[d])], chosen[d]))
    return out
