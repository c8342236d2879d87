//! Safe adaptor over `hbmpc_sys` (rust/hbmpc_sys.rs, generated from include/hbmpc_hip.h): the reference's own
//! `SecretSharingScheme<Fr>` surface (mpc/src/common/mod.rs:101-128) served by the MI355X library.
//!
//! Where it goes in the reference tree: `mpc/src/gpu/{hbmpc_sys.rs, mod.rs = this file}` behind a cargo feature
//! `gpu`; `build.rs` adds `cargo:rustc-link-search=native=$HBMPC_HIP_LIB_DIR` and `cargo:rustc-link-lib=dylib=hbmpc_hip`.
//! There is no Rust toolchain in the build image of the library, so this file has never met rustc: it is written
//! against the reference's types as they stand (file:line cited at every use) and checked textually against the
//! generated binding (tests/test_rust_shim.py: every `hbmpc_*` call below exists there with that many arguments).
//!
//! * `GpuShares`            one library context; batched calls (`compute_shares_batch`, `batch_recover_secret`, ...)
//! * `GpuRobustShare`       newtype over `RobustShare<Fr>` whose `SecretSharingScheme<Fr>` impl delegates the two
//!                          scheme functions to the process-wide `GpuShares` and every operator to the inner share
//! * `GpuPipeline`          a device-resident pipeline of ALL n simulated parties (`hbmpc_pipe_*`: triple generation,
//!                          fixed-point multiplication, RanSha, RanDouSha, run_preprocessing's triple part) -- what
//!                          `run_preprocessing` (honeybadger/mod.rs:1239-1413), `TripleGenNode::init_batch`
//!                          (triple_gen/triple_generation.rs:304-364) and `FPMulNode::init` (fpmul/fpmul.rs:61-110) drive
//! * `check`                `ShareErrorCode` -> `InterpolateError` / `ShareError`, the inverse of the mapping the
//!                          reference's C bindings apply (mpc/src/ffi/c_bindings/share/mod.rs:18-37)
use std::ffi::{CStr, CString};
use std::ops::{Add, Mul, Sub};
use std::sync::{Mutex, OnceLock};

use ark_bls12_381::Fr;
use ark_ff::{BigInt, PrimeField};
use ark_poly::{univariate::DensePolynomial, DenseUVPolynomial};
use ark_std::rand::Rng;

use super::hbmpc_sys as sys;
use crate::common::share::ShareError; // mpc/src/common/share/mod.rs:12-27
use crate::common::SecretSharingScheme; // mpc/src/common/mod.rs:101-128
use crate::honeybadger::robust_interpolate::robust_interpolate::RobustShare; // robust_interpolate.rs:17-29
use crate::honeybadger::robust_interpolate::InterpolateError; // robust_interpolate/mod.rs:7-27

// ---- Fr <-> U256: the conversion of mpc/src/ffi/c_bindings/mod.rs:37-49 -------------------------------------------
#[inline]
pub fn to_u256(x: &Fr) -> sys::U256 {
    sys::U256 { data: x.into_bigint().0 }
}
#[inline]
pub fn from_u256(x: &sys::U256) -> Fr {
    // the library only ever returns canonical values (< r), for which from_bigint is Some
    Fr::from_bigint(BigInt::new(x.data)).expect("libhbmpc_hip returned a non-canonical element")
}

/// One library context (one HIP stream).  Calls on one context serialise; create one per worker thread or share it
/// behind the mutex of `global()`.
pub struct GpuShares {
    ctx: *mut sys::HbmpcCtx,
}
// the context is internally locked (include/hbmpc_hip.h: "thread-safe and re-entrant")
unsafe impl Send for GpuShares {}
unsafe impl Sync for GpuShares {}

impl Drop for GpuShares {
    fn drop(&mut self) {
        unsafe { sys::hbmpc_destroy(self.ctx) }
    }
}

impl GpuShares {
    pub fn new(device: i32) -> Result<Self, InterpolateError> {
        let mut ctx: *mut sys::HbmpcCtx = std::ptr::null_mut();
        let rc = unsafe { sys::hbmpc_create(device, sys::Bls12_381Fr, &mut ctx) };
        if rc != sys::ShareSuccess {
            return Err(InterpolateError::InvalidInput(format!("hbmpc_create(device {device}) -> {rc}: no MI355X available")));
        }
        Ok(GpuShares { ctx })
    }

    fn last_error(&self) -> String {
        let p = unsafe { sys::hbmpc_last_error(self.ctx) };
        if p.is_null() {
            String::new()
        } else {
            unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
        }
    }

    /// `ShareErrorCode` -> the error the reference function would have returned.  `n` fills `NoSuitableDomain(n)`.
    fn check(&self, rc: sys::ShareErrorCode, n: usize) -> Result<(), InterpolateError> {
        match rc {
            sys::ShareSuccess => Ok(()),
            sys::InsufficientShares => Err(InterpolateError::ShareError(ShareError::InsufficientShares)),
            sys::DegreeMismatch => Err(InterpolateError::ShareError(ShareError::DegreeMismatch)),
            sys::IdMismatch => Err(InterpolateError::ShareError(ShareError::IdMismatch)),
            sys::TypeMismatch => Err(InterpolateError::ShareError(ShareError::TypeMismatch)),
            sys::InvalidInput => Err(InterpolateError::InvalidInput(self.last_error())),
            sys::NoSuitableDomain => Err(InterpolateError::NoSuitableDomain(n)),
            sys::PolynomialOperationError => Err(InterpolateError::PolynomialOperationError(self.last_error())),
            sys::DecodingError => Err(InterpolateError::DecodingError(self.last_error())),
            other => Err(InterpolateError::InvalidInput(format!("libhbmpc_hip error {other}: {}", self.last_error()))),
        }
    }

    /// `B` calls of `RobustShare::compute_shares` (robust_interpolate.rs:52-82) in one launch.  Draws the rng exactly
    /// like the reference does per secret: `DensePolynomial::rand(degree, rng)`, then `coeffs[0] = secret`.
    /// Returns `[party j][secret b]`.
    pub fn compute_shares_batch(&self, secrets: &[Fr], n: usize, degree: usize, rng: &mut impl Rng)
        -> Result<Vec<Vec<RobustShare<Fr>>>, InterpolateError> {
        let b = secrets.len();
        let mut coeffs = Vec::<sys::U256>::with_capacity(b * (degree + 1));
        for s in secrets {
            let mut poly = DensePolynomial::<Fr>::rand(degree, rng); // same draws, same order (:68)
            poly.coeffs[0] = *s; // (:69)
            coeffs.extend(poly.coeffs.iter().map(to_u256));
        }
        let mut out = vec![sys::U256::default(); n * b];
        let rc = unsafe { sys::hbmpc_compute_shares(self.ctx, coeffs.as_ptr(), b, n, degree, out.as_mut_ptr()) };
        self.check(rc, n)?;
        Ok((0..n)
            .map(|j| out[j * b..(j + 1) * b].iter().map(|v| RobustShare::new(from_u256(v), j, degree)).collect())
            .collect())
    }

    /// Dealer variant with the random coefficients drawn on the device ("hbmpc-chacha20-v1", include/hbmpc_hip.h).
    /// The (seed, index) pair selects the coefficients: NEVER reuse a pair for different secrets -- identical
    /// coefficients would reveal the difference of the secrets to every party.  This adaptor therefore owns the
    /// index: it is a counter of the context that only grows.
    pub fn compute_shares_seeded(&self, secrets: &[Fr], n: usize, degree: usize, seed: &[u8; 32], counter: &mut u64)
        -> Result<Vec<Vec<RobustShare<Fr>>>, InterpolateError> {
        let b = secrets.len();
        let sec: Vec<sys::U256> = secrets.iter().map(to_u256).collect();
        let mut out = vec![sys::U256::default(); n * b];
        let first_index = *counter;
        *counter = counter.checked_add(b as u64).ok_or_else(|| InterpolateError::InvalidInput("seed exhausted".into()))?;
        let rc = unsafe {
            sys::hbmpc_compute_shares_seeded(self.ctx, seed.as_ptr(), sec.as_ptr(), b, first_index, n, degree, out.as_mut_ptr())
        };
        self.check(rc, n)?;
        Ok((0..n)
            .map(|j| out[j * b..(j + 1) * b].iter().map(|v| RobustShare::new(from_u256(v), j, degree)).collect())
            .collect())
    }

    /// `RobustShare::recover_secret` (robust_interpolate.rs:94-157): validation, optimistic interpolation and the
    /// OEC/Gao fallback all happen in the library, in the reference's order and with its error codes.
    pub fn recover_secret(&self, shares: &[RobustShare<Fr>], n: usize, t: usize) -> Result<(Vec<Fr>, Fr), InterpolateError> {
        let s = shares.len();
        let ids: Vec<usize> = shares.iter().map(|x| x.id).collect();
        let degrees: Vec<usize> = shares.iter().map(|x| x.degree).collect();
        let vals: Vec<sys::U256> = shares.iter().map(|x| to_u256(&x.share[0])).collect();
        let cap = shares.first().map_or(1, |x| x.degree + 1);
        let mut coeffs = vec![sys::U256::default(); cap.max(1)];
        let mut ncoeffs: usize = 0;
        let mut secret = sys::U256::default();
        let rc = unsafe {
            sys::hbmpc_recover_secret(self.ctx, ids.as_ptr(), degrees.as_ptr(), vals.as_ptr(), s, n, t, coeffs.as_mut_ptr(),
                                      &mut ncoeffs, &mut secret)
        };
        self.check(rc, n)?;
        Ok((coeffs[..ncoeffs].iter().map(from_u256).collect(), from_u256(&secret)))
    }

    /// `batch_recover_secret` (robust_interpolate.rs:284-443): same arguments, same `Vec<Vec<F>>` (degree + 1
    /// coefficients on the optimistic path, the trimmed vector on the fallback path, the error of the lowest failing
    /// chunk as `Err`).
    pub fn batch_recover_secret(&self, evals_by_sender: &[(usize, Vec<Fr>)], n: usize, degree: usize, t: usize)
        -> Result<Vec<Vec<Fr>>, InterpolateError> {
        let s = evals_by_sender.len();
        let g = evals_by_sender.first().map_or(0, |e| e.1.len());
        if !evals_by_sender.iter().all(|(_, v)| v.len() == g) {
            return Err(InterpolateError::InvalidInput("Inconsistent batch widths".into())); // (:300-306)
        }
        let ids: Vec<usize> = evals_by_sender.iter().map(|(id, _)| *id).collect();
        let flat: Vec<sys::U256> = evals_by_sender.iter().flat_map(|(_, v)| v.iter().map(to_u256)).collect();
        let m = degree + 1;
        let mut coeffs = vec![sys::U256::default(); g * m];
        let mut ncoeffs = vec![0u32; g];
        let rc = unsafe {
            sys::hbmpc_batch_recover(self.ctx, ids.as_ptr(), s, flat.as_ptr(), g, n, degree, t, coeffs.as_mut_ptr(),
                                     ncoeffs.as_mut_ptr(), std::ptr::null_mut())
        };
        self.check(rc, n)?;
        Ok((0..g).map(|c| coeffs[c * m..c * m + ncoeffs[c] as usize].iter().map(from_u256).collect()).collect())
    }

    /// the EvalBatch arm of BatchRecon keeps coefficient 0 only (batch_recon.rs:384-391)
    pub fn batch_recover_p0(&self, evals_by_sender: &[(usize, Vec<Fr>)], n: usize, degree: usize, t: usize)
        -> Result<Vec<Fr>, InterpolateError> {
        let s = evals_by_sender.len();
        let g = evals_by_sender.first().map_or(0, |e| e.1.len());
        if !evals_by_sender.iter().all(|(_, v)| v.len() == g) {
            return Err(InterpolateError::InvalidInput("Inconsistent batch widths".into()));
        }
        let ids: Vec<usize> = evals_by_sender.iter().map(|(id, _)| *id).collect();
        let flat: Vec<sys::U256> = evals_by_sender.iter().flat_map(|(_, v)| v.iter().map(to_u256)).collect();
        let mut secrets = vec![sys::U256::default(); g];
        let rc = unsafe {
            sys::hbmpc_batch_recover_p0(self.ctx, ids.as_ptr(), s, flat.as_ptr(), g, n, degree, t, secrets.as_mut_ptr(),
                                        std::ptr::null_mut())
        };
        self.check(rc, n)?;
        Ok(secrets.iter().map(from_u256).collect())
    }

    /// `make_vandermonde` + `apply_vandermonde` over G chunks (common/share/mod.rs:31-76, batch_recon.rs:157-165):
    /// `chunks[g]` holds d + 1 values; returns `[recipient j][chunk g]`, i.e. row j is the payload for party j.
    pub fn vandermonde_apply(&self, chunks: &[Vec<Fr>], n: usize, degree: usize) -> Result<Vec<Vec<Fr>>, InterpolateError> {
        let g = chunks.len();
        if !chunks.iter().all(|c| c.len() == degree + 1) {
            return Err(InterpolateError::ShareError(ShareError::InvalidInput)); // share/mod.rs:59-64
        }
        let flat: Vec<sys::U256> = chunks.iter().flat_map(|c| c.iter().map(to_u256)).collect();
        let mut out = vec![sys::U256::default(); n * g];
        let rc = unsafe { sys::hbmpc_vandermonde_apply(self.ctx, flat.as_ptr(), g, n, degree, out.as_mut_ptr()) };
        self.check(rc, n)?;
        Ok((0..n).map(|j| out[j * g..(j + 1) * g].iter().map(from_u256).collect()).collect())
    }
}


// ---- device-resident pipelines (include/hbmpc_hip.h, "device-resident pipelines") -------------------------------------
/// One `hbmpc_pipe` handle: the call sequencing, the arena layout and the graph-capture rules live in the library.
/// Buffers are `[party][..]` arrays reached by name (`"a"`, `"b"`, `"r2t"`, `"rt"`, `"c"` for triple generation; see the
/// header for the other kinds).  `run` only enqueues on the handle's stream; `run_checked` reads the decode summaries back
/// and returns the error the reference's `?` would.
pub struct GpuPipeline<'a> {
    gpu: &'a GpuShares,
    pipe: *mut sys::HbmpcPipe,
    owned: bool,
}

impl Drop for GpuPipeline<'_> {
    fn drop(&mut self) {
        if self.owned {
            unsafe { sys::hbmpc_pipe_destroy(self.pipe) }
        }
    }
}

impl GpuShares {
    /// a HIP stream of this context's device for the pipelines (graph capture needs a real stream)
    pub fn stream_create(&self) -> Result<*mut core::ffi::c_void, InterpolateError> {
        let mut s = std::ptr::null_mut();
        let rc = unsafe { sys::hbmpc_stream_create(self.ctx, &mut s) };
        self.check(rc, 0)?;
        Ok(s)
    }
    /// A host that hands the library buffers from `hipMallocAsync`: the release threshold of the device's stream-ordered pool (0, the
    /// platform default, is the configuration under which such buffers are NOT safe: include/hbmpc_hip.h, "Device buffers") ...
    pub fn stream_pool_release_threshold(&self) -> Result<u64, InterpolateError> {
        let mut thr = 0u64;
        let rc = unsafe { sys::hbmpc_stream_pool_release_threshold(self.ctx, &mut thr) };
        self.check(rc, 0)?;
        Ok(thr)
    }
    /// ... and the one-call remedy: the pool keeps its freed blocks from here on
    pub fn stream_pool_retain(&self) -> Result<(), InterpolateError> {
        let rc = unsafe { sys::hbmpc_stream_pool_retain(self.ctx) };
        self.check(rc, 0)
    }
    fn wrap(&self, rc: sys::ShareErrorCode, pipe: *mut sys::HbmpcPipe, n: usize) -> Result<GpuPipeline<'_>, InterpolateError> {
        self.check(rc, n)?;
        Ok(GpuPipeline { gpu: self, pipe, owned: true })
    }
    /// `TripleGenNode::init_batch` + BatchRecon(2t) + finalize for all n parties (triple_gen/triple_generation.rs:304-364,164-232)
    pub fn pipe_triplegen(&self, n: usize, t: usize, triples: usize, stream: *mut core::ffi::c_void) -> Result<GpuPipeline<'_>, InterpolateError> {
        let mut p = std::ptr::null_mut();
        let rc = unsafe { sys::hbmpc_pipe_triplegen_create(self.ctx, n, t, triples, stream, &mut p) };
        self.wrap(rc, p, n)
    }
    /// `FPMulNode::init` = Beaver multiplication + TruncPr for all n parties (fpmul/fpmul.rs:61-110, fpmul/truncpr.rs:185-318)
    pub fn pipe_fpmul(&self, n: usize, t: usize, pairs: usize, k: usize, m: usize, stream: *mut core::ffi::c_void) -> Result<GpuPipeline<'_>, InterpolateError> {
        let mut p = std::ptr::null_mut();
        let rc = unsafe { sys::hbmpc_pipe_fpmul_create(self.ctx, n, t, pairs, k, m, 0, stream, &mut p) };
        self.wrap(rc, p, n)
    }
    /// `RanShaNode` for all n parties, `batch` elements per dealer (share_gen/share_gen.rs:232-289,401-454,516-530)
    pub fn pipe_ransha(&self, n: usize, t: usize, batch: usize, stream: *mut core::ffi::c_void) -> Result<GpuPipeline<'_>, InterpolateError> {
        let mut p = std::ptr::null_mut();
        let rc = unsafe { sys::hbmpc_pipe_ransha_create(self.ctx, n, t, batch, 0, stream, &mut p) };
        self.wrap(rc, p, n)
    }
    /// `DouShaNode` + `RanDouShaNode` for all n parties (ran_dou_sha/mod.rs:371-449,569-602)
    pub fn pipe_randousha(&self, n: usize, t: usize, batch: usize, stream: *mut core::ffi::c_void) -> Result<GpuPipeline<'_>, InterpolateError> {
        let mut p = std::ptr::null_mut();
        let rc = unsafe { sys::hbmpc_pipe_randousha_create(self.ctx, n, t, batch, stream, &mut p) };
        self.wrap(rc, p, n)
    }
    /// `run_preprocessing`'s triple part: RanSha -> a, b; RanDouSha -> r; TripleGen (honeybadger/mod.rs:1239-1393)
    pub fn pipe_preprocessing(&self, n: usize, t: usize, triples: usize, stream: *mut core::ffi::c_void) -> Result<GpuPipeline<'_>, InterpolateError> {
        let mut p = std::ptr::null_mut();
        let rc = unsafe { sys::hbmpc_pipe_preprocessing_create(self.ctx, n, t, triples, stream, &mut p) };
        self.wrap(rc, p, n)
    }
}

impl<'a> GpuPipeline<'a> {
    fn name(name: &str) -> CString {
        CString::new(name).expect("buffer names have no NUL")
    }
    /// a part of a preprocessing pipeline (`"ransha"`, `"randousha"`, `"triplegen"`): borrowed, lives as long as `self`
    pub fn part(&self, name: &str) -> Result<GpuPipeline<'a>, InterpolateError> {
        let mut p = std::ptr::null_mut();
        let rc = unsafe { sys::hbmpc_pipe_part(self.pipe, Self::name(name).as_ptr(), &mut p) };
        self.gpu.check(rc, 0)?;
        Ok(GpuPipeline { gpu: self.gpu, pipe: p, owned: false })
    }
    /// device pointer and element count of a named buffer (for hosts that fill it with their own device calls)
    pub fn buffer(&self, name: &str) -> Result<(*mut sys::U256, usize), InterpolateError> {
        let (mut p, mut n) = (std::ptr::null_mut(), 0usize);
        let rc = unsafe { sys::hbmpc_pipe_buffer(self.pipe, Self::name(name).as_ptr(), &mut p, &mut n) };
        self.gpu.check(rc, 0)?;
        Ok((p as *mut sys::U256, n))
    }
    /// `[party][..]` values into a named input buffer
    pub fn upload(&self, name: &str, values: &[Fr]) -> Result<(), InterpolateError> {
        let v: Vec<sys::U256> = values.iter().map(to_u256).collect();
        let rc = unsafe { sys::hbmpc_pipe_upload(self.pipe, Self::name(name).as_ptr(), v.as_ptr() as *const core::ffi::c_void, v.len()) };
        self.gpu.check(rc, 0)
    }
    /// the first `elements` values of a named buffer (synchronises)
    pub fn download(&self, name: &str, elements: usize) -> Result<Vec<Fr>, InterpolateError> {
        let mut v = vec![sys::U256::default(); elements];
        let rc = unsafe { sys::hbmpc_pipe_download(self.pipe, Self::name(name).as_ptr(), v.as_mut_ptr() as *mut core::ffi::c_void, elements) };
        self.gpu.check(rc, 0)?;
        Ok(v.iter().map(from_u256).collect())
    }
    /// enqueue the whole pipeline on its stream
    pub fn run(&self) -> Result<(), InterpolateError> {
        let rc = unsafe { sys::hbmpc_pipe_set_checked(self.pipe, 0) };
        self.gpu.check(rc, 0)?;
        let rc = unsafe { sys::hbmpc_pipe_run(self.pipe) };
        self.gpu.check(rc, 0)
    }
    /// run and read every decode's summary back: a chunk that fails ends the run with its error, as the reference's `?` does
    pub fn run_checked(&self) -> Result<(), InterpolateError> {
        let rc = unsafe { sys::hbmpc_pipe_set_checked(self.pipe, 1) };
        self.gpu.check(rc, 0)?;
        let rc = unsafe { sys::hbmpc_pipe_run(self.pipe) };
        self.gpu.check(rc, 0)
    }
    /// the two halves of a producer's run: the dealers' `compute_shares`, then everything after their messages have arrived
    pub fn deal(&self) -> Result<(), InterpolateError> {
        let rc = unsafe { sys::hbmpc_pipe_deal(self.pipe) };
        self.gpu.check(rc, 0)
    }
    pub fn finish(&self) -> Result<(), InterpolateError> {
        let rc = unsafe { sys::hbmpc_pipe_finish(self.pipe) };
        self.gpu.check(rc, 0)
    }
    /// record the call sequence as a HIP graph (after two eager runs); `replay` launches the recording
    pub fn capture(&self) -> Result<(), InterpolateError> {
        let rc = unsafe { sys::hbmpc_pipe_capture(self.pipe) };
        self.gpu.check(rc, 0)
    }
    pub fn replay(&self) -> Result<(), InterpolateError> {
        let rc = unsafe { sys::hbmpc_pipe_replay(self.pipe) };
        self.gpu.check(rc, 0)
    }
    pub fn sync(&self) -> Result<(), InterpolateError> {
        let rc = unsafe { sys::hbmpc_pipe_sync(self.pipe) };
        self.gpu.check(rc, 0)
    }
    /// the last decode's summary (synchronises)
    pub fn summary(&self) -> Result<sys::RecoverSummary, InterpolateError> {
        let mut s = sys::RecoverSummary::default();
        let rc = unsafe { sys::hbmpc_pipe_summary(self.pipe, &mut s) };
        self.gpu.check(rc, 0)?;
        Ok(s)
    }
    /// a producer's verdict: (verifier checks that failed, first failing batch element); (0, _) = every verifier says OK
    /// -- the `Ok` / abort decision of share_gen.rs:516-530 and ran_dou_sha/mod.rs:586-602 for the whole batch
    pub fn verdict(&self) -> Result<(u32, u32), InterpolateError> {
        let mut v = [0u32; 2];
        let rc = unsafe { sys::hbmpc_pipe_verdict(self.pipe, v.as_mut_ptr()) };
        self.gpu.check(rc, 0)?;
        Ok((v[0], v[1]))
    }
}

/// the process-wide context the trait impl below uses (device from HBMPC_DEVICE, default 0)
pub fn global() -> &'static Mutex<GpuShares> {
    static G: OnceLock<Mutex<GpuShares>> = OnceLock::new();
    G.get_or_init(|| {
        let dev = std::env::var("HBMPC_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
        Mutex::new(GpuShares::new(dev).expect("libhbmpc_hip: no device"))
    })
}

/// `RobustShare<Fr>` whose scheme functions run on the GPU.  Everything else -- the share record, the operators with
/// their `IdMismatch` / `DegreeMismatch` checks (common/mod.rs:167-300) -- is the reference's own code on the inner
/// share, so `HoneyBadgerMPCNode<Fr, R>` and the sub-protocol nodes compile unchanged with this type in place of
/// `RobustShare<Fr>`.
#[derive(Clone, Debug, PartialEq)]
pub struct GpuRobustShare(pub RobustShare<Fr>);

impl From<RobustShare<Fr>> for GpuRobustShare {
    fn from(s: RobustShare<Fr>) -> Self {
        GpuRobustShare(s)
    }
}
impl Add for GpuRobustShare {
    type Output = Result<Self, ShareError>;
    fn add(self, rhs: Self) -> Self::Output {
        (self.0 + rhs.0).map(GpuRobustShare)
    }
}
impl Sub for GpuRobustShare {
    type Output = Result<Self, ShareError>;
    fn sub(self, rhs: Self) -> Self::Output {
        (self.0 - rhs.0).map(GpuRobustShare)
    }
}
impl Add<Fr> for GpuRobustShare {
    type Output = Result<Self, ShareError>;
    fn add(self, rhs: Fr) -> Self::Output {
        (self.0 + rhs).map(GpuRobustShare)
    }
}
impl Sub<Fr> for GpuRobustShare {
    type Output = Result<Self, ShareError>;
    fn sub(self, rhs: Fr) -> Self::Output {
        (self.0 - rhs).map(GpuRobustShare)
    }
}
impl Mul<Fr> for GpuRobustShare {
    type Output = Result<Self, ShareError>;
    fn mul(self, rhs: Fr) -> Self::Output {
        (self.0 * rhs).map(GpuRobustShare)
    }
}

impl SecretSharingScheme<Fr> for GpuRobustShare {
    type SecretType = Fr;
    type Error = InterpolateError;

    /// one secret = a batch of one (`ids` is ignored, as in the reference: robust_interpolate.rs:56)
    fn compute_shares(secret: Fr, n: usize, degree: usize, _ids: Option<&[usize]>, rng: &mut impl Rng)
        -> Result<Vec<Self>, InterpolateError> {
        let gpu = global().lock().expect("hbmpc context poisoned");
        let by_party = gpu.compute_shares_batch(&[secret], n, degree, rng)?;
        Ok(by_party.into_iter().map(|mut row| GpuRobustShare(row.remove(0))).collect())
    }

    fn recover_secret(shares: &[Self], n: usize, t: usize) -> Result<(Vec<Fr>, Fr), InterpolateError> {
        let inner: Vec<RobustShare<Fr>> = shares.iter().map(|s| s.0.clone()).collect();
        global().lock().expect("hbmpc context poisoned").recover_secret(&inner, n, t)
    }
}
