### Additive R entry points over the batch routines of libpeaksegdisk_hip.so.  The five
### entry points of the package (PeakSegFPOP_file/_dir/_df/_vec, sequentialSearch_dir) keep
### working unchanged through .C("PeakSegFPOP_interface"); these are optional faster forms.

PeakSegFPOP_dir_batch <- function
### PeakSegFPOP_dir for many (problem.dir, penalty) pairs: cached results are reused, the rest
### is solved in one launch (one parse and upload per coverage.bedGraph).
(problem.dir.vec, penalty.vec){
  stopifnot(is.character(problem.dir.vec), length(problem.dir.vec)==length(penalty.vec))
  n <- length(problem.dir.vec)
  res <- .C(
    "PeakSegFPOP_dir_batch_interface",
    as.character(problem.dir.vec), paste(penalty.vec), as.integer(n),
    status=integer(n), cached=integer(n),
    PACKAGE="PeakSegDisk")
  if(any(res$status != 0)){
    stop("error code ", res$status[res$status != 0][1])
  }
  ## every pair now has consistent result files: these calls are cache hits
  mapply(PeakSegFPOP_dir, problem.dir.vec, paste(penalty.vec), SIMPLIFY=FALSE)
}

sequentialSearch_dir_resident <- function
### sequentialSearch_dir with the coverage parsed and uploaded once and the arena reused
### between penalties; same penalties, same files, same result as sequentialSearch_dir.
(problem.dir, peaks.int, verbose=0){
  stopifnot(is.integer(peaks.int) && length(peaks.int)==1 && 0 <= peaks.int)
  stopifnot(is.character(problem.dir) && length(problem.dir)==1)
  capacity <- 256L
  res <- .C(
    "PeakSegFPOP_search_interface",
    problem.dir, peaks.int, as.integer(verbose), capacity,
    penalty=rep(strrep(" ", 39), capacity), iteration=integer(capacity),
    under=integer(capacity), over=integer(capacity),
    n=integer(1), chosen=integer(1),
    PACKAGE="PeakSegDisk")
  i.vec <- seq_len(res$n)
  model.list <- lapply(i.vec, function(i){
    L <- PeakSegFPOP_dir(problem.dir, res$penalty[i])
    L$loss$iteration <- res$iteration[i]
    L$loss$under <- res$under[i]
    L$loss$over <- res$over[i]
    L
  })
  out <- model.list[[res$chosen]]
  out$others <- do.call(rbind, lapply(model.list, "[[", "loss"))[order(iteration)]
  out
}

sequentialSearch_dir_batch <- function
### sequentialSearch_dir on several problem directories at once: every directory gets the
### result sequentialSearch_dir(problem.dir, peaks.int) gives, but the models the searches ask
### for in the same iteration are computed in one launch on the GPU.
(problem.dir.vec, peaks.int.vec, verbose=0){
  stopifnot(is.character(problem.dir.vec), is.integer(peaks.int.vec), all(0 <= peaks.int.vec))
  n <- length(problem.dir.vec)
  peaks.int.vec <- rep(peaks.int.vec, l=n)
  capacity <- 256L
  res <- .C(
    "PeakSegFPOP_search_batch_interface",
    problem.dir.vec, as.integer(n), peaks.int.vec, as.integer(verbose), capacity,
    penalty=rep(strrep(" ", 39), capacity*n), iteration=integer(capacity*n),
    under=integer(capacity*n), over=integer(capacity*n),
    n.models=integer(n), chosen=integer(n), status=integer(n),
    PACKAGE="PeakSegDisk")
  if(any(res$status != 0)){
    stop("error code ", res$status[res$status != 0][1])
  }
  lapply(seq_len(n), function(d){
    i.vec <- (d-1L)*capacity + seq_len(res$n.models[d])
    model.list <- lapply(i.vec, function(i){
      L <- PeakSegFPOP_dir(problem.dir.vec[d], res$penalty[i])
      L$loss$iteration <- res$iteration[i]
      L$loss$under <- res$under[i]
      L$loss$over <- res$over[i]
      L
    })
    out <- model.list[[res$chosen[d]]]
    out$others <- do.call(rbind, lapply(model.list, "[[", "loss"))[order(iteration)]
    out
  })
}
