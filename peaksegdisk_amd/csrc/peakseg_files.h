/* peakseg_files.h -- the file-level layers of the host driver (included by peakseg_hip.cpp).
 *
 *   PeakSegFPOP_disk / _disk_batch   the reference's native boundary
 *                                    (/root/reference/src/PeakSegFPOPLog.cpp:143-463, "drv")
 *   PeakSegFPOP_dir_batch            PeakSegFPOP_dir's cache protocol and _timing.tsv next to the
 *                                    batch entry (/root/reference/R/PeakSegFPOP_dir.R:64-117,
 *                                    R/PeakSegFPOP_file.R:30-87)
 *   PeakSegFPOP_sequential_search    sequentialSearch_dir's loop with the contig parsed and
 *                                    uploaded once and the arena reused
 *                                    (/root/reference/R/sequentialSearch_dir.R:22-103; f2)
 *
 * Every dynamic program runs on the GPU (peakseg_hip_problem_set_*); this file only parses,
 * formats and moves files.
 */
#include <limits.h>
#include <sys/stat.h>

namespace {

double wall_now() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* R's paste()/as.character() of a double: 15 significant digits, trailing zeros dropped, the
 * narrower of fixed and scientific notation (R's formatReal with digits = 15).  The penalties
 * of sequentialSearch_dir reach the solver and the file names through this
 * (R/sequentialSearch_dir.R:43, R/PeakSegFPOP_dir.R:64), and write.table() formats
 * _timing.tsv the same way (R/PeakSegFPOP_dir.R:102-106). */
std::string r_paste_double(double x) {
  if (x != x) return "NaN";
  if (std::isinf(x)) return x > 0 ? "Inf" : "-Inf";
  if (x == 0) return "0";
  char buf[64];
  snprintf(buf, sizeof buf, "%.14e", x);
  const double target = strtod(buf, nullptr);
  int nsig = 15;
  for (int d = 1; d <= 15; d++) { /* fewest digits that reproduce the 15-digit value */
    snprintf(buf, sizeof buf, "%.*e", d - 1, x);
    if (strtod(buf, nullptr) == target) {
      nsig = d;
      break;
    }
  }
  snprintf(buf, sizeof buf, "%.*e", nsig - 1, x);
  const char *e = strchr(buf, 'e');
  const int kpower = e ? atoi(e + 1) : 0;
  const int neg = x < 0 ? 1 : 0;
  int left, rgt;
  if (kpower >= 0) {
    left = kpower + 1;
    rgt = nsig - kpower - 1;
    if (rgt < 0) rgt = 0;
  } else {
    left = 1;
    rgt = nsig - kpower - 1;
  }
  const int w_fixed = neg + left + (rgt > 0 ? rgt + 1 : 0);
  const int w_sci = neg + (nsig > 1 ? nsig + 1 : 1) + (abs(kpower) >= 100 ? 5 : 4);
  char out[400];
  if (w_fixed <= w_sci) {
    snprintf(out, sizeof out, "%.*f", rgt, x);
  } else {
    snprintf(out, sizeof out, "%.*e", nsig - 1, x);
  }
  return out;
}

/* ---- one (bedGraph, penalty, db) problem of the file-level boundary -------------------- */

struct FileProblem {
  const char *bedGraph = nullptr, *penalty_str = nullptr, *db = nullptr;
  int status = 0;
  bool is_Inf = false;
  double penalty = 0.0;
  int cov = -1;      /* index into the parsed-coverage table */
  int dp_index = -1; /* index into the device problem set, -1 = trivial/none */
  bool outputs_opened = false;
  bool loss_failed = false, segments_failed = false;
  /* what the loss file says (kept for PeakSegFPOP_dir_batch / the penalty search) */
  int n_segments = 0, n_peaks = 0, bases = 0;
  double total_loss = 0.0;
  long long db_bytes = 0; /* size of the reference's cost-function database for this problem */
};

std::string out_prefix(const FileProblem &fp) {
  return std::string(fp.bedGraph) + "_penalty=" + fp.penalty_str;
}

/* create / truncate, as the reference's ofstream::open does at drv:212-223 */
bool truncate_file(const std::string &path) {
  FILE *f = fopen(path.c_str(), "w");
  if (!f) return false;
  return fclose(f) == 0;
}

bool write_whole_file(const std::string &path, const std::string &text) {
  FILE *f = fopen(path.c_str(), "w");
  if (!f) return false;
  bool ok = text.empty() || fwrite(text.data(), 1, text.size(), f) == text.size();
  if (fclose(f) != 0) ok = false;
  return ok;
}

void append_fmt(std::string &s, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  int n = vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (n > 0) s.append(buf, (size_t)(n < (int)sizeof buf ? n : (int)sizeof buf - 1));
}

/* Output files are created before the trivial/DP split (drv:212-223) and each of them is
 * written in one piece later: no descriptor stays open across the kernel (a batch of several
 * hundred problems used to hold two FILE* each, past the usual 1024-descriptor limit). */
void open_outputs(FileProblem &fp) {
  const std::string pre = out_prefix(fp);
  fp.loss_failed = !truncate_file(pre + "_loss.tsv");
  fp.segments_failed = !truncate_file(pre + "_segments.bed");
  fp.outputs_opened = true;
}

void write_outputs(FileProblem &fp, const std::string &segments, const std::string &loss) {
  const std::string pre = out_prefix(fp);
  if (!fp.loss_failed && !write_whole_file(pre + "_loss.tsv", loss)) fp.loss_failed = true;
  if (!fp.segments_failed && !write_whole_file(pre + "_segments.bed", segments))
    fp.segments_failed = true;
}

/* trivial one-segment model (drv:224-243) */
void write_trivial(FileProblem &fp, const Coverage &cv) {
  double best_cost;
  if (cv.cum_weighted_count != 0) {
    best_cost =
        cv.cum_weighted_count * (1 - psd_log(cv.cum_weighted_count) + psd_log(cv.cum_weight));
  } else {
    best_cost = 0;
  }
  std::string seg, loss;
  append_fmt(seg, "%s\t%d\t%d\tbackground\t%g\n", cv.chrom.c_str(), cv.first_chromStart,
             cv.chromEnd.back(), cv.cum_weighted_count / cv.cum_weight);
  append_fmt(loss, "%s\t%d\t%d\t%d\t%d\t%.20g\t%.20g\t%d\t%d\t%d\n", fp.penalty_str, 1, 0,
             (int)cv.cum_weight, cv.n(), best_cost / cv.cum_weight, best_cost, 0, 0, 0);
  write_outputs(fp, seg, loss);
  fp.n_segments = 1;
  fp.n_peaks = 0;
  fp.bases = (int)cv.cum_weight;
  fp.total_loss = best_cost;
  fp.db_bytes = 0;
}

/* one problem's results, copied off the device */
struct DpFetched {
  int status = 0;
  psd_result r{};
  std::vector<int> seg_start;
  std::vector<double> seg_mean;
};

void fetch_dp(int dp_index, psd_problem_set *set, DpFetched &f) {
  if (peakseg_hip_problem_set_result(set, dp_index, &f.r) != 0 || f.r.status != 0) {
    f.status = ERROR_DEVICE_SOLVER;
    return;
  }
  f.seg_start.resize((size_t)f.r.n_segments);
  f.seg_mean.resize((size_t)f.r.n_segments);
  if (peakseg_hip_problem_set_segments(set, dp_index, f.r.n_segments, f.seg_start.data(),
                                       f.seg_mean.data()) != f.r.n_segments)
    f.status = ERROR_DEVICE_SOLVER;
}

/* the DP branch's two files (drv:419-454) */
int write_dp_outputs(FileProblem &fp, const Coverage &cv, const DpFetched &f) {
  if (f.status) return f.status;
  const psd_result &r = f.r;
  int prev_chromEnd = cv.chromEnd.back();
  const char *chrom = cv.chrom.c_str();
  std::string seg, loss;
  seg.reserve((size_t)r.n_segments * 48);
  for (int row = 0; row < r.n_segments; row++) {
    int start = f.seg_start[(size_t)row] < 0 ? cv.first_chromStart
                                             : cv.chromEnd[(size_t)f.seg_start[(size_t)row]];
    /* rows alternate background/peak starting and ending with background (drv:421-429,442) */
    const char *status_str = (row % 2 == 0) ? "background" : "peak";
    append_fmt(seg, "%s\t%d\t%d\t%s\t%g\n", chrom, start, prev_chromEnd, status_str,
               f.seg_mean[(size_t)row]);
    prev_chromEnd = start;
  }
  int n_peaks = (r.n_segments - 1) / 2;
  double total_intervals = (double)r.total_intervals;
  const double total_loss = r.best_cost * cv.cum_weight - fp.penalty * n_peaks;
  append_fmt(loss, "%.20g\t%d\t%d\t%d\t%d\t%.20g\t%.20g\t%d\t%.20g\t%.20g\n", fp.penalty,
             r.n_segments, n_peaks, (int)cv.cum_weight, cv.n(), r.best_cost, total_loss,
             r.n_equality_constraints, total_intervals / (cv.n() * 2), (double)r.max_intervals);
  write_outputs(fp, seg, loss);
  fp.n_segments = r.n_segments;
  fp.n_peaks = n_peaks;
  fp.bases = (int)cv.cum_weight;
  fp.total_loss = total_loss;
  /* leave a sparse file of the size the reference's DiskVector would have:
   * 2N 16-byte positions + per function {int size, int n, int chromEnd} + 20 B per piece */
  fp.db_bytes = 32ll * cv.n() + 12ll * (2ll * cv.n() - 1) + 20ll * (long long)r.total_intervals;
  if (truncate(fp.db, (off_t)fp.db_bytes) != 0) return ERROR_WRITING_COST_FUNCTIONS;
  return 0;
}

/* write failures are reported at the end, loss first (drv:456-461) */
void settle_status(FileProblem &fp) {
  if (fp.status == 0 && fp.outputs_opened) {
    if (fp.loss_failed) {
      fp.status = ERROR_WRITING_LOSS_OUTPUT;
    } else if (fp.segments_failed) {
      fp.status = ERROR_WRITING_SEGMENTS_OUTPUT;
    }
  }
}

/* the DP branch touches the cost-function database first (drv:247-252) */
bool touch_db(FileProblem &fp) {
  FILE *db = fopen(fp.db, "w+b");
  if (!db) {
    fp.status = ERROR_WRITING_COST_FUNCTIONS;
    return false;
  }
  fclose(db);
  return true;
}

std::string real_path(const std::string &path) {
  char buf[PATH_MAX];
  if (realpath(path.c_str(), buf)) return buf;
  return path;
}

int solve_files(int n, FileProblem *fps) {
  /* PEAKSEG_HIP_TIMING=1: where a call spends its time, on stderr */
  const bool timing = getenv("PEAKSEG_HIP_TIMING") != nullptr;
  double t_mark = wall_now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    double now = wall_now();
    fprintf(stderr, "peakseg_hip timing: %-28s %8.3f s\n", what, now - t_mark);
    t_mark = now;
  };
  std::vector<Coverage> covs;
  std::map<std::string, int> cov_of_path;
  std::map<std::string, int> cov_status;
  /* A (bedGraph, penalty string) pair listed twice names the same two output files -- the
   * reference builds the names from the path string as given (drv:212-223) -- so the second copy
   * is not processed at all (its writer thread would race the first one's on the same paths);
   * it receives the first one's status and figures at the end.  Two DIFFERENT strings that reach
   * the same file (a symlinked coverage.bedGraph, as in PeakSegPipeline problem directories)
   * name different output files: each entry gets its own, but the file is parsed once and the
   * dynamic program of a (file, penalty) pair runs once (step 3). */
  std::vector<int> dup_of((size_t)n, -1);
  {
    std::map<std::pair<std::string, std::string>, int> seen;
    for (int i = 0; i < n; i++) {
      auto key = std::make_pair(std::string(fps[i].bedGraph), std::string(fps[i].penalty_str));
      auto it = seen.find(key);
      if (it == seen.end()) {
        seen[key] = i;
      } else {
        dup_of[(size_t)i] = it->second;
      }
    }
  }
  /* 1. penalties, then inputs (validation order of drv:145-209) */
  for (int i = 0; i < n; i++) {
    FileProblem &fp = fps[i];
    if (dup_of[(size_t)i] >= 0) continue;
    fp.status = parse_penalty(fp.penalty_str, fp.is_Inf, fp.penalty);
    if (fp.status) continue;
    const std::string path = real_path(fp.bedGraph); /* one parse per file, whatever its names */
    auto it = cov_of_path.find(path);
    if (it == cov_of_path.end()) {
      Coverage cv;
      int st = read_bedGraph(fp.bedGraph, cv);
      covs.push_back(std::move(cv));
      int idx = (int)covs.size() - 1;
      cov_of_path[path] = idx;
      cov_status[path] = st;
      it = cov_of_path.find(path);
    }
    fp.cov = it->second;
    fp.status = cov_status[path];
  }
  lap("parse bedGraph");
  /* 2. output files are created before the trivial/DP split (drv:212-223); the db is only
   *    touched in the DP branch (drv:247-252) */
  std::vector<int> dp;
  for (int i = 0; i < n; i++) {
    FileProblem &fp = fps[i];
    if (fp.status || dup_of[(size_t)i] >= 0) continue;
    const Coverage &cv = covs[(size_t)fp.cov];
    open_outputs(fp);
    if (fp.is_Inf || cv.min_log_mean == cv.max_log_mean) {
      write_trivial(fp, cv);
    } else if (touch_db(fp)) {
      dp.push_back(i);
    }
  }
  /* 3. all dynamic programs in one device problem set */
  if (!dp.empty()) {
    std::vector<int> contig_of_cov(covs.size(), -1);
    std::vector<int> contig_n, prob_contig;
    std::vector<const int *> cnt_ptr, wt_ptr;
    std::vector<double> prob_pen;
    std::map<std::pair<int, std::string>, int> dp_of; /* (file, penalty string) -> device problem */
    for (int i : dp) {
      FileProblem &fp = fps[i];
      if (contig_of_cov[(size_t)fp.cov] < 0) {
        contig_of_cov[(size_t)fp.cov] = (int)contig_n.size();
        const Coverage &cv = covs[(size_t)fp.cov];
        contig_n.push_back(cv.n());
        cnt_ptr.push_back(cv.count.data());
        wt_ptr.push_back(cv.weight.data());
      }
      auto key = std::make_pair(fp.cov, std::string(fp.penalty_str));
      auto it = dp_of.find(key);
      if (it != dp_of.end()) { /* the same file under another name: its problem is there already */
        fp.dp_index = it->second;
        continue;
      }
      fp.dp_index = (int)prob_contig.size();
      dp_of[key] = fp.dp_index;
      prob_contig.push_back(contig_of_cov[(size_t)fp.cov]);
      prob_pen.push_back(fp.penalty);
    }
    psd_problem_set *set = nullptr;
    int st = peakseg_hip_problem_set_create(env_device(), (int)contig_n.size(), contig_n.data(),
                                            cnt_ptr.data(), wt_ptr.data(), (int)prob_contig.size(),
                                            prob_contig.data(), prob_pen.data(), 0, &set);
    lap("upload + allocate");
    if (st == 0) {
      st = peakseg_hip_problem_set_solve(set, nullptr, nullptr);
      if (st == ERROR_DEVICE_SOLVER) st = 0; /* per-problem statuses decide below */
    }
    lap("kernel");
    /* results leave the device one problem after the other; the text files (the segment
     * tables of a penalty grid are hundreds of MB) are then formatted by a few threads */
    std::vector<DpFetched> fetched(prob_contig.size());
    for (size_t k = 0; k < fetched.size(); k++) {
      if (st) {
        fetched[k].status = st;
      } else {
        fetch_dp((int)k, set, fetched[k]);
      }
    }
    if (set) peakseg_hip_problem_set_destroy(set);
    lap("download results + free");
    std::atomic<size_t> next_k{0};
    auto writer = [&]() {
      for (size_t k = next_k++; k < dp.size(); k = next_k++) {
        FileProblem &fp = fps[dp[k]];
        fp.status = write_dp_outputs(fp, covs[(size_t)fp.cov], fetched[(size_t)fp.dp_index]);
      }
    };
    unsigned n_threads = std::thread::hardware_concurrency();
    if (n_threads > 16) n_threads = 16;
    if (n_threads > dp.size()) n_threads = (unsigned)dp.size();
    if (n_threads <= 1) {
      writer();
    } else {
      std::vector<std::thread> pool;
      for (unsigned w = 0; w < n_threads; w++) pool.emplace_back(writer);
      for (auto &th : pool) th.join();
    }
    lap("write segments/loss files");
  }
  /* 4. report write failures (drv:456-461: loss first) */
  int first = 0;
  for (int i = 0; i < n; i++) {
    FileProblem &fp = fps[i];
    if (dup_of[(size_t)i] >= 0) continue;
    settle_status(fp);
  }
  for (int i = 0; i < n; i++) {
    FileProblem &fp = fps[i];
    if (dup_of[(size_t)i] >= 0) {
      const FileProblem &o = fps[dup_of[(size_t)i]];
      fp.status = o.status;
      fp.n_segments = o.n_segments;
      fp.n_peaks = o.n_peaks;
      fp.bases = o.bases;
      fp.total_loss = o.total_loss;
      fp.db_bytes = o.db_bytes;
      /* its own database name, if it has one, is left as the first copy's is */
      if (fp.status == 0 && o.dp_index >= 0 && strcmp(fp.db, o.db) != 0 && touch_db(fp) &&
          truncate(fp.db, (off_t)fp.db_bytes) != 0)
        fp.status = ERROR_WRITING_COST_FUNCTIONS;
    }
    if (fp.status && !first) first = fp.status;
  }
  return first;
}

/* ---- PeakSegFPOP_dir's result-file cache (R/PeakSegFPOP_dir.R:70-93) -------------------- */

bool read_small_file(const std::string &path, std::string &out) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char chunk[4096];
  size_t got;
  out.clear();
  while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) {
    out.append(chunk, got);
    if (out.size() > (1u << 20)) break; /* one-row files */
  }
  fclose(f);
  return true;
}

std::vector<std::string> split_fields(const std::string &line) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < line.size()) {
    while (i < line.size() && (line[i] == '\t' || line[i] == ' ' || line[i] == '\r')) i++;
    size_t j = i;
    while (j < line.size() && line[j] != '\t' && line[j] != ' ' && line[j] != '\r') j++;
    if (j > i) out.push_back(line.substr(i, j - i));
    i = j;
  }
  return out;
}

std::vector<std::string> nonempty_lines(const std::string &text) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < text.size()) {
    size_t j = text.find('\n', i);
    if (j == std::string::npos) j = text.size();
    std::string line = text.substr(i, j - i);
    if (!split_fields(line).empty()) out.push_back(line);
    i = j + 1;
  }
  return out;
}

/* first and last line of a (possibly large) text file without reading all of it */
bool first_last_line(const std::string &path, std::string &first, std::string &last) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[8192];
  bool ok = fgets(buf, sizeof buf, f) != nullptr;
  if (ok) {
    first = buf;
    while (!first.empty() && (first.back() == '\n' || first.back() == '\r')) first.pop_back();
    ok = fseek(f, 0, SEEK_END) == 0;
  }
  if (ok) {
    long size = ftell(f);
    long back = size < (long)sizeof buf - 1 ? size : (long)sizeof buf - 1;
    ok = fseek(f, size - back, SEEK_SET) == 0;
    if (ok) {
      size_t got = fread(buf, 1, (size_t)back, f);
      std::string tail(buf, got);
      std::vector<std::string> lines = nonempty_lines(tail);
      ok = !lines.empty();
      if (ok) last = lines.back();
    }
  }
  fclose(f);
  return ok && !split_fields(first).empty();
}

bool parse_int_field(const std::string &s, long long &v) {
  char *end = nullptr;
  errno = 0;
  v = strtoll(s.c_str(), &end, 10);
  return end != s.c_str() && *end == 0 && errno == 0;
}

bool parse_double_field(const std::string &s, double &v) {
  char *end = nullptr;
  v = strtod(s.c_str(), &end);
  return end != s.c_str() && *end == 0;
}

struct LossRow { /* the columns of _loss.tsv the callers use (R/col.name.list.R:12-15) */
  double penalty = 0.0, total_loss = 0.0;
  long long segments = 0, peaks = 0, bases = 0;
};

/* TRUE when the three result files of (problem dir, penalty) exist and are consistent, as
 * PeakSegFPOP_dir decides before it reuses them; any failure means recompute. */
bool dir_cache_ok(const std::string &bedGraph, const std::string &pre, LossRow &row) {
  std::string text, first_seg, last_seg, first_cov, last_cov;
  if (!read_small_file(pre + "_timing.tsv", text)) return false;
  std::vector<std::string> tl = nonempty_lines(text);
  if (tl.size() != 1 || split_fields(tl[0]).size() != 3) return false;
  if (!first_last_line(pre + "_segments.bed", first_seg, last_seg)) return false;
  if (!first_last_line(bedGraph, first_cov, last_cov)) return false;
  if (!read_small_file(pre + "_loss.tsv", text)) return false;
  std::vector<std::string> ll = nonempty_lines(text);
  if (ll.size() != 1) return false;
  std::vector<std::string> lf = split_fields(ll[0]);
  std::vector<std::string> fs = split_fields(first_seg), ls = split_fields(last_seg);
  std::vector<std::string> fc = split_fields(first_cov), lc = split_fields(last_cov);
  if (lf.size() != 10 || fs.size() != 5 || ls.size() != 5 || fc.size() != 4 || lc.size() != 4)
    return false;
  long long fs_end, ls_start, fc_start, lc_end;
  if (!parse_int_field(fs[2], fs_end) || !parse_int_field(ls[1], ls_start) ||
      !parse_int_field(fc[1], fc_start) || !parse_int_field(lc[2], lc_end))
    return false;
  if (!parse_double_field(lf[0], row.penalty) || !parse_int_field(lf[1], row.segments) ||
      !parse_int_field(lf[2], row.peaks) || !parse_int_field(lf[3], row.bases) ||
      !parse_double_field(lf[6], row.total_loss))
    return false;
  return fs_end - ls_start == row.bases && fc_start == ls_start && lc_end == fs_end;
}

/* _timing.tsv: penalty, megabytes, seconds the way write.table() prints them
 * (R/PeakSegFPOP_dir.R:98-106) */
bool write_timing(const std::string &pre, const char *penalty_str, double megabytes,
                  double seconds) {
  std::string t = r_paste_double(strtod(penalty_str, nullptr)) + "\t" +
                  r_paste_double(megabytes) + "\t" + r_paste_double(seconds) + "\n";
  return write_whole_file(pre + "_timing.tsv", t);
}

bool file_exists(const std::string &path) {
  struct stat st;
  return stat(path.c_str(), &st) == 0;
}

/* ---- a problem directory kept resident for a sequence of penalties ---------------------- */

struct ResidentDir {
  std::string dir, bedGraph, norm;
  bool parsed = false;
  int parse_status = 0;
  Coverage cv;
  psd_problem_set *set = nullptr;
  double kernel_s = 0.0;
  int solves = 0;

  ~ResidentDir() {
    if (set) peakseg_hip_problem_set_destroy(set);
  }

  /* PeakSegFPOP_dir(problem.dir, penalty.str) (R/PeakSegFPOP_dir.R:64-117 over
   * R/PeakSegFPOP_file.R:57-86): reuse consistent result files, else solve, then write
   * _timing.tsv.  The contig is parsed and uploaded on the first dynamic program only. */
  int model(const char *pen_str, LossRow &row, bool &cached) {
    const std::string pre = bedGraph + "_penalty=" + pen_str;
    cached = dir_cache_ok(bedGraph, pre, row);
    if (cached) return 0;
    const double t0 = wall_now();
    FileProblem fp;
    const std::string db = norm + "_penalty=" + pen_str + ".db";
    fp.bedGraph = norm.c_str();
    fp.penalty_str = pen_str;
    fp.db = db.c_str();
    unlink(db.c_str());
    fp.status = parse_penalty(pen_str, fp.is_Inf, fp.penalty);
    if (fp.status) return fp.status;
    if (!parsed) {
      parse_status = read_bedGraph(norm.c_str(), cv);
      parsed = true;
    }
    if (parse_status) return parse_status;
    open_outputs(fp);
    if (fp.is_Inf || cv.min_log_mean == cv.max_log_mean) {
      write_trivial(fp, cv);
    } else if (touch_db(fp)) {
      int st = 0;
      if (!set) {
        const int n = cv.n();
        const int *cnt = cv.count.data(), *wt = cv.weight.data();
        const int contig = 0;
        st = peakseg_hip_problem_set_create(env_device(), 1, &n, &cnt, &wt, 1, &contig,
                                            &fp.penalty, 0, &set);
      } else {
        st = peakseg_hip_problem_set_set_penalty(set, 0, fp.penalty) == 0 ? 0 : ERROR_DEVICE_SOLVER;
      }
      DpFetched f;
      if (st == 0) {
        float ms = 0.f;
        st = peakseg_hip_problem_set_solve(set, &ms, nullptr);
        kernel_s += ms / 1e3;
        solves++;
      }
      if (st) {
        f.status = st;
      } else {
        fetch_dp(0, set, f);
      }
      fp.status = write_dp_outputs(fp, cv, f);
    }
    settle_status(fp);
    const double megabytes = file_exists(db) ? (double)fp.db_bytes / 1024.0 / 1024.0 : 0.0;
    unlink(db.c_str());
    if (fp.status) return fp.status;
    if (!write_timing(pre, pen_str, megabytes, wall_now() - t0)) return ERROR_WRITING_LOSS_OUTPUT;
    row.penalty = fp.penalty;
    row.segments = fp.n_segments;
    row.peaks = fp.n_peaks;
    row.bases = fp.bases;
    /* the callers read total.loss back from the 20-digit text of _loss.tsv: same double */
    row.total_loss = fp.total_loss;
    return 0;
  }
};

}  // namespace

extern "C" int PeakSegFPOP_disk(char *bedGraph_file_name, char *penalty_str, char *db_file_name) {
  FileProblem fp;
  fp.bedGraph = bedGraph_file_name;
  fp.penalty_str = penalty_str;
  fp.db = db_file_name;
  return solve_files(1, &fp);
}

extern "C" int PeakSegFPOP_disk_batch(int n_problems, char **bedGraph_files, char **penalty_strs,
                                      char **db_files, int *status_out) {
  if (n_problems <= 0) return 0;
  std::vector<FileProblem> fps((size_t)n_problems);
  for (int i = 0; i < n_problems; i++) {
    fps[(size_t)i].bedGraph = bedGraph_files[i];
    fps[(size_t)i].penalty_str = penalty_strs[i];
    fps[(size_t)i].db = db_files[i];
  }
  int first = solve_files(n_problems, fps.data());
  if (status_out)
    for (int i = 0; i < n_problems; i++) status_out[i] = fps[(size_t)i].status;
  return first;
}

extern "C" int PeakSegFPOP_dir_batch(int n_problems, char **problem_dirs, char **penalty_strs,
                                     int *status_out, int *cached_out) {
  if (n_problems <= 0) return 0;
  const double t0 = wall_now();
  std::vector<std::string> bedGraph((size_t)n_problems), norm((size_t)n_problems),
      db((size_t)n_problems);
  std::vector<FileProblem> fps;
  std::vector<int> todo;
  int first = 0;
  for (int i = 0; i < n_problems; i++) {
    bedGraph[(size_t)i] = std::string(problem_dirs[i]) + "/coverage.bedGraph";
    const std::string pre = bedGraph[(size_t)i] + "_penalty=" + penalty_strs[i];
    LossRow row;
    const bool hit = dir_cache_ok(bedGraph[(size_t)i], pre, row);
    if (cached_out) cached_out[i] = hit ? 1 : 0;
    if (status_out) status_out[i] = 0;
    if (hit) continue;
    /* PeakSegFPOP_file: normalised path, default db name, db removed before the call */
    norm[(size_t)i] = real_path(bedGraph[(size_t)i]);
    db[(size_t)i] = norm[(size_t)i] + "_penalty=" + penalty_strs[i] + ".db";
    unlink(db[(size_t)i].c_str());
    todo.push_back(i);
  }
  fps.resize(todo.size());
  for (size_t k = 0; k < todo.size(); k++) {
    const size_t i = (size_t)todo[k];
    fps[k].bedGraph = norm[i].c_str();
    fps[k].penalty_str = penalty_strs[i];
    fps[k].db = db[i].c_str();
  }
  if (!todo.empty()) solve_files((int)fps.size(), fps.data());
  /* seconds: the reference times each call on its own; here the problems of a batch run
   * concurrently, so each one is charged the batch's wall time in proportion to its data */
  const double wall = wall_now() - t0;
  double bins_total = 0.0;
  for (auto &fp : fps) bins_total += fp.status == 0 ? (double)fp.bases : 0.0;
  for (size_t k = 0; k < todo.size(); k++) {
    const size_t i = (size_t)todo[k];
    FileProblem &fp = fps[k];
    const double megabytes = file_exists(db[i]) ? (double)fp.db_bytes / 1024.0 / 1024.0 : 0.0;
    unlink(db[i].c_str());
    if (fp.status == 0) {
      const std::string pre = bedGraph[i] + "_penalty=" + penalty_strs[i];
      const double seconds = bins_total > 0 ? wall * (double)fp.bases / bins_total : wall;
      if (!write_timing(pre, penalty_strs[i], megabytes, seconds))
        fp.status = ERROR_WRITING_LOSS_OUTPUT;
    }
    if (status_out) status_out[i] = fp.status;
    if (fp.status && !first) first = fp.status;
  }
  return first;
}

namespace {

/* The decisions of sequentialSearch_dir (R/sequentialSearch_dir.R:39-99) for one problem
 * directory, apart from how a model is computed: the single-directory entry computes its models
 * one after the other on a resident contig, the batch entry computes the pending model of every
 * directory in one device launch.  Same state machine, so the same sequence of penalties. */
struct SearchState {
  int peaks_int = 0, row_capacity = 0;
  psd_search_row *rows = nullptr;
  int n = 0, under = -1, over = -1, candidate = -1, iteration = 0, first_new = 0;
  int status = 0;
  std::vector<double> next_pen = {0.0, INFINITY};

  bool active() const { return status == 0 && !next_pen.empty(); }
  void begin_iteration() {
    iteration++;
    first_new = n;
  }
  /* the next row, its penalty string filled in; nullptr when the table is full */
  psd_search_row *new_row(double pen) {
    if (n >= row_capacity) {
      set_error("sequential search: more than %d models", row_capacity);
      status = ERROR_SEARCH_ARGUMENTS;
      return nullptr;
    }
    psd_search_row &r = rows[n];
    memset(&r, 0, sizeof r);
    const std::string pen_str = r_paste_double(pen);
    snprintf(r.penalty_str, sizeof r.penalty_str, "%s", pen_str.c_str());
    return &r;
  }
  void record(psd_search_row &r, const LossRow &lr, bool cached) {
    const int NA = INT_MIN;
    r.iteration = iteration;
    r.under_peaks = under < 0 ? NA : rows[under].peaks;
    r.over_peaks = over < 0 ? NA : rows[over].peaks;
    r.penalty = lr.penalty;
    r.peaks = (int)lr.peaks;
    r.segments = (int)lr.segments;
    r.bases = (int)lr.bases;
    r.total_loss = lr.total_loss;
    r.cached = cached ? 1 : 0;
    n++;
  }
  /* after the models of this iteration: the new bracket and the next penalty */
  void end_iteration() {
    if (iteration == 1) {
      over = first_new;      /* penalty 0 */
      under = first_new + 1; /* penalty Inf */
      const int max_peaks = (rows[over].bases - 1) / 2;
      if (max_peaks < peaks_int) {
        set_error("peaks.int=%d but max=%d peaks for N=%d data", peaks_int, max_peaks,
                  rows[over].bases);
        status = ERROR_SEARCH_TOO_MANY_PEAKS;
        return;
      }
    } else {
      const int m = first_new;
      if (rows[m].peaks == rows[under].peaks || rows[m].peaks == rows[over].peaks) {
        candidate = under; /* not a new model: pick the simpler one */
        next_pen.clear();
      } else if (rows[m].peaks < peaks_int) {
        under = m;
      } else {
        over = m;
      }
    }
    if (peaks_int == rows[under].peaks) {
      candidate = under;
      next_pen.clear();
    }
    if (peaks_int == rows[over].peaks) {
      candidate = over;
      next_pen.clear();
    }
    if (!next_pen.empty()) {
      const double pen = (rows[over].total_loss - rows[under].total_loss) /
                         (double)(rows[under].peaks - rows[over].peaks);
      next_pen.clear();
      if (pen < 0) {
        candidate = under; /* numerically unstable region: return the simpler model */
      } else {
        next_pen.push_back(pen);
      }
    }
  }
};

void search_say_next(const std::vector<double> &next_pen) {
  std::string line = "Next =";
  for (size_t k = 0; k < next_pen.size(); k++)
    line += (k ? ", " : " ") + r_paste_double(next_pen[k]);
  emit_text("%s \n", line.c_str());
}

}  // namespace

extern "C" int PeakSegFPOP_sequential_search(const char *problem_dir, int peaks_int, int verbose,
                                             int row_capacity, psd_search_row *rows, int *n_rows,
                                             int *chosen_row) {
  if (n_rows) *n_rows = 0;
  if (chosen_row) *chosen_row = -1;
  if (!problem_dir || peaks_int < 0 || !rows || row_capacity < 2) {
    set_error("sequential search: bad arguments");
    return ERROR_SEARCH_ARGUMENTS;
  }
  ResidentDir rd;
  rd.dir = problem_dir;
  rd.bedGraph = rd.dir + "/coverage.bedGraph";
  rd.norm = real_path(rd.bedGraph);
  SearchState ss;
  ss.peaks_int = peaks_int;
  ss.row_capacity = row_capacity;
  ss.rows = rows;
  while (ss.active()) {
    if (verbose) search_say_next(ss.next_pen);
    ss.begin_iteration();
    const std::vector<double> pens = ss.next_pen;
    for (double pen : pens) {
      psd_search_row *r = ss.new_row(pen);
      if (!r) return ss.status;
      LossRow lr;
      bool cached = false;
      int st = rd.model(r->penalty_str, lr, cached);
      if (st) {
        if (n_rows) *n_rows = ss.n;
        return st;
      }
      ss.record(*r, lr, cached);
      if (getenv("PEAKSEG_HIP_TIMING")) /* progress of a long search (stderr is unbuffered) */
        fprintf(stderr, "peakseg_hip timing: search model %d: penalty=%s peaks=%d%s, %.1f s so far "
                        "in the kernel\n", ss.n, r->penalty_str, r->peaks, cached ? " (cached)" : "",
                rd.kernel_s);
    }
    if (n_rows) *n_rows = ss.n;
    ss.end_iteration();
    if (ss.status) return ss.status;
  }
  if (chosen_row) *chosen_row = ss.candidate;
  if (getenv("PEAKSEG_HIP_TIMING"))
    fprintf(stderr, "peakseg_hip timing: sequential search: %d models, %d dynamic programs, "
                    "%.3f s in the kernel\n", ss.n, rd.solves, rd.kernel_s);
  return 0;
}

/* sequentialSearch_dir over several problem directories at once (additive entry): every
 * directory follows its own search, exactly as PeakSegFPOP_sequential_search would, but the
 * models the searches ask for in the same iteration are computed in ONE launch
 * (PeakSegFPOP_dir_batch: one problem per directory, the chip shared between them).  One
 * search keeps four wave slots of 8192 busy; a genome's worth of contigs searched together
 * costs about what its longest search costs.  rows: n_dirs x row_capacity; n_rows, chosen_row,
 * status_out: per directory.  Returns the first non-zero status (0: every search ended). */
extern "C" int PeakSegFPOP_sequential_search_batch(int n_dirs, char **problem_dirs,
                                                   const int *peaks_int, int verbose,
                                                   int row_capacity, psd_search_row *rows,
                                                   int *n_rows, int *chosen_row,
                                                   int *status_out) {
  if (n_dirs <= 0) return 0;
  if (!problem_dirs || !peaks_int || !rows || row_capacity < 2) {
    set_error("sequential search: bad arguments");
    return ERROR_SEARCH_ARGUMENTS;
  }
  std::vector<SearchState> ss((size_t)n_dirs);
  std::vector<std::string> bedGraph((size_t)n_dirs);
  for (int d = 0; d < n_dirs; d++) {
    ss[(size_t)d].peaks_int = peaks_int[d];
    ss[(size_t)d].row_capacity = row_capacity;
    ss[(size_t)d].rows = rows + (size_t)d * (size_t)row_capacity;
    if (peaks_int[d] < 0 || !problem_dirs[d]) ss[(size_t)d].status = ERROR_SEARCH_ARGUMENTS;
    if (problem_dirs[d]) bedGraph[(size_t)d] = std::string(problem_dirs[d]) + "/coverage.bedGraph";
    if (n_rows) n_rows[d] = 0;
    if (chosen_row) chosen_row[d] = -1;
  }
  /* a directory listed twice would have two searches write the same files in the same launch */
  for (int d = 0; d < n_dirs; d++)
    for (int e = 0; e < d; e++)
      if (ss[(size_t)d].status == 0 && real_path(bedGraph[(size_t)d]) == real_path(bedGraph[(size_t)e])) {
        set_error("sequential search: problem directory %s is listed twice", problem_dirs[d]);
        ss[(size_t)d].status = ERROR_SEARCH_ARGUMENTS;
      }
  const double t0 = wall_now();
  int round = 0, launches = 0;
  for (;;) {
    /* the models wanted now: (directory, row) pairs; a directory's first iteration asks for
     * two (penalties 0 and Inf), later ones for one */
    std::vector<int> who;
    std::vector<psd_search_row *> row_of;
    for (int d = 0; d < n_dirs; d++) {
      SearchState &s = ss[(size_t)d];
      if (!s.active()) continue;
      if (verbose) {
        emit_text("%s: ", problem_dirs[d]);
        search_say_next(s.next_pen);
      }
      s.begin_iteration();
      const std::vector<double> pens = s.next_pen;
      for (double pen : pens) {
        psd_search_row *r = s.new_row(pen);
        if (!r) break;
        who.push_back(d);
        row_of.push_back(r);
        s.n++; /* reserved; filled in below */
      }
      if (s.status) continue;
      s.n = s.first_new; /* record() counts them again */
    }
    if (who.empty()) break;
    round++;
    /* drop the models of directories that have just failed */
    std::vector<char *> dirs, pens;
    std::vector<size_t> slot;
    for (size_t k = 0; k < who.size(); k++) {
      if (ss[(size_t)who[k]].status) continue;
      dirs.push_back(problem_dirs[who[k]]);
      pens.push_back(row_of[k]->penalty_str);
      slot.push_back(k);
    }
    std::vector<int> st(dirs.size(), 0), cached(dirs.size(), 0);
    if (!dirs.empty()) {
      PeakSegFPOP_dir_batch((int)dirs.size(), dirs.data(), pens.data(), st.data(), cached.data());
      launches++;
    }
    for (size_t j = 0; j < slot.size(); j++) {
      const size_t k = slot[j];
      const int d = who[k];
      SearchState &s = ss[(size_t)d];
      if (s.status) continue;
      if (st[j]) {
        s.status = st[j];
        continue;
      }
      LossRow lr;
      const std::string pre = bedGraph[(size_t)d] + "_penalty=" + row_of[k]->penalty_str;
      if (!dir_cache_ok(bedGraph[(size_t)d], pre, lr)) {
        set_error("sequential search: result files of %s are not consistent", pre.c_str());
        s.status = ERROR_DEVICE_SOLVER;
        continue;
      }
      s.record(*row_of[k], lr, cached[j] != 0);
    }
    for (int d = 0; d < n_dirs; d++) {
      SearchState &s = ss[(size_t)d];
      if (s.status || s.n == s.first_new || s.iteration == 0) continue;
      bool mine = false;
      for (size_t k = 0; k < who.size(); k++) mine = mine || who[k] == d;
      if (!mine) continue;
      if (n_rows) n_rows[d] = s.n;
      s.end_iteration();
    }
    if (getenv("PEAKSEG_HIP_TIMING"))
      fprintf(stderr, "peakseg_hip timing: search batch round %d: %zu models, %.1f s so far\n",
              round, dirs.size(), wall_now() - t0);
  }
  int first = 0;
  for (int d = 0; d < n_dirs; d++) {
    const SearchState &s = ss[(size_t)d];
    if (n_rows) n_rows[d] = s.n;
    if (chosen_row) chosen_row[d] = s.status ? -1 : s.candidate;
    if (status_out) status_out[d] = s.status;
    if (s.status && !first) first = s.status;
  }
  if (getenv("PEAKSEG_HIP_TIMING"))
    fprintf(stderr, "peakseg_hip timing: sequential search batch: %d directories, %d rounds, "
                    "%d launches, %.3f s\n", n_dirs, round, launches, wall_now() - t0);
  return first;
}

/* Tests: R's paste() of a double as this library formats penalties and timing files. */
extern "C" int peakseg_hip_paste_double(double x, char *buf, size_t buf_len) {
  std::string s = r_paste_double(x);
  if (!buf || buf_len == 0) return (int)s.size();
  snprintf(buf, buf_len, "%s", s.c_str());
  return (int)s.size();
}

