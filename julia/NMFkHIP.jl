# NMFkHIP.jl -- Julia host side of libnmfk_hip.so (include/nmfk_hip.h): the drop-in for NMFk.execute(...; method=:simple).
#
#     import NMFkHIP
#     W, H, fitquality, robustness, aic, kopt = NMFkHIP.execute(X, 2:16, 32; ngpus=8)
#
# Written to offer the same methods, keyword arguments, return shapes, result files and exception types as the reference path
# (src/NMFkExecute.jl:15-65, 178-233, 236-329, 483-711, 729-807; src/NMFkMultiplicative.jl:24; citations below are
# relative to the NMFk.jl source tree).  What differs is where the work happens: the reference's serial loops over k
# (Exec:203) and over the restarts (Exec:535-541, or pmap Exec:511-526) become ONE flat (k, restart) work list that
# libnmfk_hip runs on 1..8 MI355X (nmfk_mu_sweep / nmfk_multi_sweep), followed by the robustness step per k on GPU 0.
#
# STATUS: written against the C ABI and checked statically (tests/test_host_cpu.py: every ccall against the header and
# the exported symbols; every keyword of the reference signatures present; block structure balanced; the per-restart seed
# rule and the order of the post-processing steps diffed against the Python mirror).  It has NOT run: the build container
# has no `julia`.  The Python mirror nmfk.jl_amd/execute.py implements the same orchestration and is what the GPU tests run.
module NMFkHIP

import Random
import Statistics
import LinearAlgebra
import SparseArrays
import SHA
import Serialization
import JLD
import Printf

const libnmfk = get(ENV, "NMFK_HIP_LIB", joinpath(@__DIR__, "..", "nmfk.jl_amd", "libnmfk_hip.so"))
global_quiet = true
first_warning = true  # Mult:8-15: the zero row / column warnings appear once per session

# ---------------------------------------------------------------------------------------------------------------
# C ABI
# ---------------------------------------------------------------------------------------------------------------
# nmfk_mu_params (include/nmfk_hip.h) == keyword arguments of NMFmultiplicative (Mult:24) as forwarded by Exec:729,762
Base.@kwdef mutable struct MuParams
	tol::Cdouble = 1e-19
	tolOF::Cdouble = 1e-3
	lambda::Cdouble = 1e-32
	weight::Cdouble = 1.0
	maxiter::Int64 = 10000
	maxreattempts::Int32 = 2
	maxbaditers::Int32 = 10
	stopconv::Int32 = 1000
	Wfixed::Int32 = 0
	Hfixed::Int32 = 0
	normalize::Int32 = 1
	compute::Int32 = 0
	reserved::Int32 = 0
end

"status -> the exception the reference would have thrown"
function check(rc::Integer)
	rc == 0 && return nothing
	msg = unsafe_string(ccall((:nmfk_last_error, libnmfk), Cstring, ()))
	rc == 2 && throw(ErrorException(msg))             # "All matrix entries must be nonnegative!" (Mult:4-7)
	rc == 3 && error(msg)                              # NaNs in the initial factors (Mult:42-44, 52-54)
	error("libnmfk_hip: $msg (status $rc)")
end

"One GPU, or (ngpus > 1) the GPUs 0..ngpus-1 of this node behind one handle (nmfk_multi_*: a context, an RCCL communicator
and a host thread per GPU; X is broadcast, the restarts are sharded, results come back through GPU 0)."
mutable struct Context
	h::Ptr{Cvoid}       # nmfk_ctx of GPU 0 (clustering, silhouettes, fit checks)
	multi::Ptr{Cvoid}   # nmfk_multi or C_NULL
	ngpus::Int
	function Context(; device::Integer=0, ngpus::Integer=1)
		ENV["GPU_MAX_HW_QUEUES"] = get(ENV, "GPU_MAX_HW_QUEUES", "24") # one hardware queue per rank group; before HIP starts
		r = Ref{Ptr{Cvoid}}(C_NULL)
		if ngpus > 1
			check(ccall((:nmfk_multi_create, libnmfk), Cint, (Cint, Ref{Ptr{Cvoid}}), ngpus, r))
			mh = r[]
			check(ccall((:nmfk_multi_context, libnmfk), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), mh, 0, r))
			c = new(r[], mh, ngpus)
			finalizer(x -> ccall((:nmfk_multi_destroy, libnmfk), Cint, (Ptr{Cvoid},), x.multi), c)
		else
			check(ccall((:nmfk_create, libnmfk), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r))
			c = new(r[], C_NULL, 1)
			finalizer(x -> ccall((:nmfk_destroy, libnmfk), Cint, (Ptr{Cvoid},), x.h), c)
		end
		return c
	end
end

"NMFpreprocessing! (Mult:3-22) on the device copy/copies; the caller's X is never modified.  Returns count(isnan, X)."
function setX!(c::Context, X::AbstractMatrix{<:Real}; lambda::Number=1e-32)
	Xf = convert(Matrix{Float32}, X)
	nan = Ref{Int64}(0); zero = Ref{Int64}(0)
	GC.@preserve Xf begin
		if c.multi != C_NULL
			check(ccall((:nmfk_multi_set_X, libnmfk), Cint,
				(Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Int64, Cdouble, Ref{Int64}, Ref{Int64}),
				c.multi, Xf, size(Xf, 1), size(Xf, 2), stride(Xf, 2), lambda, nan, zero))
		else
			check(ccall((:nmfk_set_X, libnmfk), Cint,
				(Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Int64, Cdouble, Ref{Int64}, Ref{Int64}),
				c.h, Xf, size(Xf, 1), size(Xf, 2), stride(Xf, 2), lambda, nan, zero))
		end
	end
	return nan[]
end

"Sparse X (BASELINE configs[3]: zeros stay zeros, which is the reference arithmetic to < 1e-30 because a zero becomes
lambda = 1e-32, Mult:17-18).  nmfk_set_X_csc takes HOST pointers and zero-based indices; with several GPUs every GPU's
context is given the matrix (there is no device-side broadcast of the CSC arrays).  Returns 0 (no missing entries: NaN
needs the dense path, the library says so)."
function setX!(c::Context, X::SparseArrays.SparseMatrixCSC{<:Real,<:Integer}; lambda::Number=1e-32)
	n, m = size(X)
	colptr = Int64.(SparseArrays.getcolptr(X)) .- Int64(1)
	rowidx = Int32.(SparseArrays.rowvals(X) .- 1)
	vals = Float32.(SparseArrays.nonzeros(X))
	kept = Ref{Int64}(0)
	GC.@preserve colptr rowidx vals begin
		for g in 0:(c.ngpus - 1)
			h = c.h
			if c.multi != C_NULL
				r = Ref{Ptr{Cvoid}}(C_NULL)
				check(ccall((:nmfk_multi_context, libnmfk), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), c.multi, g, r))
				h = r[]
			end
			check(ccall((:nmfk_set_X_csc, libnmfk), Cint,
				(Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float32}, Ref{Int64}),
				h, n, m, length(vals), colptr, rowidx, vals, kept))
		end
	end
	return 0
end

"array-valued weight of the monitored objective (Mult:74); shapes of the assertion at Exec:484"
function setweight!(c::Context, weight, n::Int, m::Int)
	if weight isa Number
		check(ccall((:nmfk_set_weight, libnmfk), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64), c.h, C_NULL, 0, 0))
		return
	end
	c.multi != C_NULL && error("array-valued weight with ngpus > 1 is not supported: use one GPU")
	wm = convert(Matrix{Float32}, ones(Float32, n, m) .* weight)
	GC.@preserve wm check(ccall((:nmfk_set_weight, libnmfk), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64), c.h, wm, n, m))
end

"All restarts of all ranks in `ks` (replaces Exec:203 x Exec:535-541).  Winit/Hinit: per rank, n x k x nNMF / k x m x nNMF."
function mu_sweep(c::Context, n::Int, m::Int, ks::Vector{Int}, nNMF::Int, p::MuParams, Wi::Vector{Array{Float32,3}},
		Hi::Vector{Array{Float32,3}})
	Wo = [Array{Float32}(undef, n, k, nNMF) for k in ks]
	Ho = [Array{Float32}(undef, k, m, nNMF) for k in ks]
	fo = [Vector{Float32}(undef, nNMF) for _ in ks]
	so = [Vector{Float64}(undef, nNMF) for _ in ks]
	io = [Vector{Int32}(undef, nNMF) for _ in ks]
	ro = [Vector{Int32}(undef, nNMF) for _ in ks]
	ptrs(v) = [pointer(a) for a in v]
	kk = Int32.(ks)
	GC.@preserve Wi Hi Wo Ho fo so io ro kk begin
		if c.multi != C_NULL
			check(ccall((:nmfk_multi_sweep, libnmfk), Cint,
				(Ptr{Cvoid}, Cint, Ptr{Int32}, Cint, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{UInt64}, Ref{MuParams},
				 Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Int32}}, Ptr{Ptr{Int32}}),
				c.multi, length(ks), kk, nNMF, ptrs(Wi), ptrs(Hi), C_NULL, p,
				ptrs(Wo), ptrs(Ho), ptrs(fo), ptrs(so), ptrs(io), ptrs(ro)))
		else
			check(ccall((:nmfk_mu_sweep, libnmfk), Cint,
				(Ptr{Cvoid}, Cint, Ptr{Int32}, Cint, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{UInt64}, Ref{MuParams},
				 Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Int32}}, Ptr{Ptr{Int32}}),
				c.h, length(ks), kk, nNMF, ptrs(Wi), ptrs(Hi), C_NULL, p,
				ptrs(Wo), ptrs(Ho), ptrs(fo), ptrs(so), ptrs(io), ptrs(ro)))
		end
	end
	return Wo, Ho, fo, so
end

"clustersolutions + silhouettes (Clus:425-517, Fin:36-66); stack: k x len x nsol, sorted by objective"
function cluster_silhouette(c::Context, stack::Array{Float32,3})
	k, m, nsol = size(stack)
	labels = Matrix{Int32}(undef, k, nsol); cent = Matrix{Float32}(undef, k, m)
	psil = Matrix{Float32}(undef, k, nsol); csil = Vector{Float32}(undef, k)
	GC.@preserve stack check(ccall((:nmfk_cluster_silhouette, libnmfk), Cint,
		(Ptr{Cvoid}, Cint, Cint, Int64, Ptr{Float32}, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
		c.h, k, nsol, m, stack, labels, cent, psil, csil))
	return Int.(labels), cent, psil, csil
end

"silhouettes for given labels (finalize on the mutated W stack of the clusterWmatrix path, Fin:45-50)"
function silhouette(c::Context, stack::Array{Float32,3}, labels::Matrix{Int})
	k, m, nsol = size(stack)
	lab = Int32.(labels); psil = Matrix{Float32}(undef, k, nsol); csil = Vector{Float32}(undef, k)
	GC.@preserve stack lab check(ccall((:nmfk_silhouette, libnmfk), Cint,
		(Ptr{Cvoid}, Cint, Cint, Int64, Ptr{Float32}, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}),
		c.h, k, nsol, m, stack, lab, psil, csil))
	return psil, csil
end

"cluster means and corrected variances (Fin:64-77); Wst: n x k x nsol, Hst: k x m x nsol"
function cluster_stats(c::Context, Wst::Array{Float32,3}, Hst::Array{Float32,3}, labels::Matrix{Int})
	n, k, nsol = size(Wst); m = size(Hst, 2)
	lab = Int32.(labels)
	Wm = Matrix{Float32}(undef, n, k); Wv = similar(Wm); Hm = Matrix{Float32}(undef, k, m); Hv = similar(Hm)
	GC.@preserve Wst Hst lab check(ccall((:nmfk_cluster_stats, libnmfk), Cint,
		(Ptr{Cvoid}, Cint, Cint, Int64, Int64, Ptr{Float32}, Ptr{Float32}, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
		c.h, k, nsol, n, m, Wst, Hst, lab, Wm, Hm, Wv, Hv))
	return Wm, Hm, Wv, Hv
end

"normnan(X - W*H) (Help:226-228)"
function frobenius(c::Context, W::AbstractMatrix, H::AbstractMatrix)
	Wf = convert(Matrix{Float32}, W); Hf = convert(Matrix{Float32}, H)
	out = Ref{Float64}(0)
	GC.@preserve Wf Hf check(ccall((:nmfk_frobenius, libnmfk), Cint, (Ptr{Cvoid}, Cint, Ptr{Float32}, Ptr{Float32}, Ref{Float64}),
		c.h, size(Wf, 2), Wf, Hf, out))
	return out[]
end

"robustkmeans(X, k, repeats) (src/NMFkCluster.jl:172-246) on the GPU; X: d x n, columns = samples"
function robustkmeans(c::Context, X::Matrix{Float32}, k::Integer, repeats::Integer=1000; maxiter::Integer=1000,
		tol::Number=1e-32, seed::Integer=0, compute_silhouettes_flag::Bool=false)
	d, n = size(X)
	assignments = Vector{Int32}(undef, n); centers = Matrix{Float32}(undef, d, k); costs = Vector{Float32}(undef, n)
	counts = Vector{Int32}(undef, k); totalcost = Ref{Float64}(0)
	best = Ref{Int32}(0); iters = Ref{Int32}(0); nclusters = Ref{Int32}(0); converged = Ref{Int32}(0)
	sil = compute_silhouettes_flag ? Vector{Float32}(undef, n) : Float32[]
	GC.@preserve X sil check(ccall((:nmfk_robustkmeans_ex, libnmfk), Cint,
		(Ptr{Cvoid}, Cint, Int64, Ptr{Float32}, Cint, Cint, Cint, Cdouble, UInt64, Ptr{Int32}, Ptr{Float32}, Ptr{Float32},
		 Ptr{Int32}, Ref{Float64}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ptr{Float64}, Ptr{Float32}, Ref{Int32}),
		c.h, d, n, X, k, repeats, maxiter, tol, UInt64(seed), assignments, centers, costs, counts, totalcost, best, iters,
		nclusters, C_NULL, compute_silhouettes_flag ? pointer(sil) : C_NULL, converged))
	nclusters[] < k && @warn("Robust k-means analysis could not find $k clusters! Only $(nclusters[]) clusters were found.")
	res = (assignments=Int.(assignments), centers=centers[:, 1:nclusters[]], costs=costs, counts=Int.(counts[1:nclusters[]]),
		totalcost=totalcost[], iterations=Int(iters[]), converged=converged[] != 0)
	return compute_silhouettes_flag ? (res, sil) : res
end

# ---------------------------------------------------------------------------------------------------------------
# host logic of the reference, unchanged in meaning
# ---------------------------------------------------------------------------------------------------------------
"getk (src/NMFkPostprocess.jl:7-41)"
function getk(nkrange::Union{AbstractRange{T1},AbstractVector{T1}}, robustness::AbstractVector{T2}, cutoff::Number=0.5; strict::Bool=true) where {T1 <: Integer, T2 <: Number}
	if length(nkrange) != length(robustness)                                          # Post:8-10: the full k-indexed vector
		robustness = robustness[nkrange]
	end
	all(isnan.(robustness)) && return 0
	if length(nkrange) == 1
		return strict ? (robustness[end] > cutoff ? nkrange[end] : nothing) : nkrange[end]
	end
	kn = findlast(i -> i > cutoff, robustness)
	if isnothing(kn)
		strict && return nothing
		rr = map(r -> isnan(r) ? -Inf : r, robustness)
		return nkrange[findmax(rr)[2]]
	end
	return nkrange[kn]
end

"signalorder (src/NMFkPostprocess.jl:148-158): sum(W[:,i:i] * H[i:i,:]) = colsum(W)_i * rowsum(H)_i"
signalorder(W::AbstractMatrix, H::AbstractMatrix) = sortperm(vec(sum(W; dims=1)) .* vec(sum(H; dims=2)); rev=true)

hash_sha256_hex(X) = (io = IOBuffer(); Serialization.serialize(io, X); bytes2hex(SHA.sha256(take!(io))))  # Exec:62-66

"check_x_hash! (Exec:68-93): the digests are the reference's own (same Serialization + SHA), so sidecars are interchangeable"
function check_x_hash!(X, xfile::AbstractString; quiet::Bool=false)
	h = hash_sha256_hex(X)
	hashfile = xfile * ".sha256"
	if isfile(hashfile)
		stored = strip(read(hashfile, String))
		if stored != "" && stored != h
			@warn("Matrix hash mismatch in '$(hashfile)': Cached results may not correspond to this matrix! Consider deleting the hash file and cached results to avoid confusion.")
		elseif !quiet
			@info("Matrix hash DOES match the stored hash in '$(hashfile)'.")
		end
	else
		mkpath(dirname(abspath(hashfile)))
		write(hashfile, h * "\n")
		!quiet && @info("Matrix hash saved in '$(hashfile)'.")
	end
	return h
end

"input_checks (Exec:95-175) for the option space of this path"
function input_checks(X::AbstractArray{T,N}, load::Bool, save::Bool, casefilename::AbstractString, mixture::Symbol, method::Symbol, algorithm::Symbol, clusterWmatrix::Bool) where {T <: Number, N}
	if (load || save) && casefilename == ""
		casefilename = "nmfk"
	end
	N > 2 && throw(ArgumentError("NMFk analysis can be executed for matrices!"))  # tensors: NMFk.jl's own path
	mixture != :null && error("mixture=$(mixture) (MixMatch) is outside the :simple path; use NMFk.execute")
	if method in (:multdiv, :multmse, :alspgrad)  # Exec:138-147
		algorithm = method
		method = :nmf
	end
	if method != :simple
		if any(isnan, X) && method in (:nmf, :sparsity)  # Exec:128-130
			@warn("Analyzed matrix has NaN's! NMF method $(method) cannot be used! Simple multiplicative NMF will be performed!")
			method = :simple
		else
			error("method=$(method) is a different solver; NMFkHIP implements method=:simple (use NMFk.execute)")
		end
	end
	return load, save, casefilename, mixture, method, algorithm, clusterWmatrix
end

"NMFk.ExecuteOptions (Exec:15-30)"
Base.@kwdef struct ExecuteOptions
	cutoff::Float64 = 0.5
	clusterWmatrix::Bool = false
	mixture::Symbol = :null
	method::Symbol = :simple
	algorithm::Symbol = :multdiv
	resultdir::String = "."
	load::Bool = true
	save::Bool = true
	casefilename::String = ""
	dims::Any = 1:2
	loadonly::Bool = false
	quiet::Bool = false
	check_inputs::Bool = true
	ordersignals::Bool = true
end

"options overload for a range of k (Exec:33-47)"
function execute(X::AbstractArray{T,N}, nkrange::Union{Vector{Int},AbstractUnitRange{Int}}, nNMF::Integer, opts::ExecuteOptions; kw...) where {T <: Number, N}
	return execute(X, nkrange, nNMF; cutoff=opts.cutoff, clusterWmatrix=opts.clusterWmatrix, mixture=opts.mixture, method=opts.method,
		algorithm=opts.algorithm, resultdir=opts.resultdir, load=opts.load, save=opts.save, casefilename=opts.casefilename, dims=opts.dims, kw...)
end

"options overload for one k (Exec:50-65)"
function execute(X::AbstractArray{T,N}, nk::Integer, nNMF::Integer, opts::ExecuteOptions; kw...) where {T <: Number, N}
	return execute(X, nk, nNMF; clusterWmatrix=opts.clusterWmatrix, mixture=opts.mixture, method=opts.method, algorithm=opts.algorithm,
		resultdir=opts.resultdir, casefilename=opts.casefilename, loadonly=opts.loadonly, load=opts.load, save=opts.save, quiet=opts.quiet,
		check_inputs=opts.check_inputs, ordersignals=opts.ordersignals, dims=opts.dims, kw...)
end

# keyword arguments of NMFmultiplicative (Mult:24) / execute_singlerun_compute (Exec:729) peeled out of kw...
const MU_KEYS = (:tol, :tolOF, :lambda, :maxiter, :maxreattempts, :maxbaditers, :stopconv, :compute)

"Initial factors of all restarts of rank nk, drawn with JULIA's RNG in the reference's order (restart ascending, W then H,
Mult:38,48), so that Random.seed!(s); execute(...) starts from the reference's points.  With the keyword `seed` given
(`seed === nothing`: not given), restart r is drawn after `Random.seed!(seed + r)` -- the reference forwards
`seed=kwseed+i` to restart i (Exec:536, 540) and NMFmultiplicative re-seeds when that sum is >= 0 (Mult:33-35): every
restart starts from DIFFERENT factors.  Given Winit / Hinit are used for every restart (Mult:40-41, 50-51)."
function draw_inits(n::Int, m::Int, nk::Int, nNMF::Int; seed::Union{Nothing,Integer}=nothing, Winit=Matrix{Float32}(undef, 0, 0), Hinit=Matrix{Float32}(undef, 0, 0))
	Wi = Array{Float32}(undef, n, nk, nNMF); Hi = Array{Float32}(undef, nk, m, nNMF)
	for r in 1:nNMF
		!isnothing(seed) && seed + r >= 0 && Random.seed!(seed + r)                   # Exec:536,540 -> Mult:33-35
		if sizeof(Winit) == 0
			Wi[:, :, r] = rand(n, nk)
		else
			@assert size(Winit) == (n, nk)
			any(isnan, Winit) && error("Initial values for the W matrix entries include NaNs!")
			Wi[:, :, r] = Winit
		end
		if sizeof(Hinit) == 0
			Hi[:, :, r] = rand(nk, m)
		else
			@assert size(Hinit) == (nk, m)
			any(isnan, Hinit) && error("Initial values for the H matrix entries include NaNs!")
			Hi[:, :, r] = Hinit
		end
	end
	return Wi, Hi
end

resultfile(resultdir, casefilename, X, nk, nNMF, suffix="") = joinpath(resultdir, "$(casefilename)_$(size(X,1))_$(size(X,2))_$(nk)_$(nNMF)$(suffix).jld")

"everything of execute_run after the restart loop (Exec:545-710) for one k; WBig/HBig/objvalue as at Exec:529-531"
function execute_run_post(c::Context, X::AbstractMatrix{T}, nk::Int, nNMF::Int, WBig::Vector{Matrix{T}}, HBig::Vector{Matrix{T}}, objvalue::Vector{T},
		wsse::Vector{Float64}, nancount::Int; clusterWmatrix::Bool=false, acceptratio::Number=1, acceptfactor::Number=Inf, quiet::Bool=true,
		veryquiet::Bool=true, best::Bool=true, resultdir::AbstractString=".", casefilename::AbstractString="", nanaction::Symbol=:zeroed,
		saveall::Bool=false) where {T <: Number}
	idxsort = sortperm(objvalue)                                                      # Exec:545
	bestIdx = idxsort[1]
	Wbest = copy(WBig[bestIdx]); Hbest = copy(HBig[bestIdx])
	if acceptratio < 1                                                                # Exec:552-558
		ccc = convert(Int, ceil(nNMF * acceptratio))
		idxrat = vec([trues(ccc); falses(nNMF - ccc)])
		@warn("NMF solutions removed based on an acceptance ratio: $(sum(idxrat)) out of $(nNMF) solutions remain")
	else
		idxrat = trues(nNMF)
	end
	if acceptfactor < Inf                                                             # Exec:559-565
		idxcut = objvalue[idxsort] .< objvalue[bestIdx] * acceptfactor
		@warn("NMF solutions removed based on an acceptance factor: $(sum(idxcut)) out of $(nNMF) solutions remain")
	else
		idxcut = trues(nNMF)
	end
	idxnan = trues(nNMF)
	if nanaction == :zeroed                                                           # Exec:567-580
		zerod = 0
		for i in idxsort
			isnw = isnan.(WBig[i]); WBig[i][isnw] .= 0
			isnh = isnan.(HBig[i]); HBig[i][isnh] .= 0
			(sum(isnw) > 0 || sum(isnh) > 0) && (zerod += 1)
		end
		zerod > 0 && @warn("NMF solutions contain NaN's: $(zerod) out of $(nNMF) solutions! NaN's have been converted to zeros!")
	elseif nanaction == :removed                                                      # Exec:581-595
		for i in idxsort
			(any(isnan, WBig[i]) || any(isnan, HBig[i])) && (idxnan[i] = false)
		end
		sum(idxnan) < nNMF && @warn("NMF solutions removed because they contain NaN's: $(sum(idxnan)) out of $(nNMF) solutions remain")
	end
	idxsol = idxrat .& idxcut .& idxnan                                               # Exec:596
	if sum(idxsol) < nNMF
		println("NMF solutions removed based on various criteria: $(sum(idxsol)) out of $(nNMF) solutions remain")
	end
	for i in 1:nNMF                                                                   # Exec:601-606: weighted residual norm = sqrt of the library's sse
		of = sqrt(max(wsse[i], 0.0))
		if of > 0 && abs(of - objvalue[i]) / of > 1e-4
			@warn("OF $i is very different: $(of) vs $(objvalue[i])!")
		end
	end
	minsilhouette = 1
	Wv = NaN; Hv = NaN
	local clustersilhouettes, clusterassignments, clustercentroids
	if nk > 1
		Ws = WBig[idxsort][idxsol]; Hs = HBig[idxsort][idxsol]
		nsol = length(Hs)
		Hst = Array{Float32}(undef, nk, size(X, 2), nsol)
		Wst = Array{Float32}(undef, size(X, 1), nk, nsol)
		for t in 1:nsol
			Hst[:, :, t] = Hs[t]; Wst[:, :, t] = Ws[t]
		end
		if clusterWmatrix                                                             # Exec:621: the W matrices themselves, in place (Clus:453-455, 484, 512)
			Wt = permutedims(Wst, (2, 1, 3))
			clusterassignments, clustercentroids, _, _ = cluster_silhouette(c, Wt)
			Ws[1] .= permutedims(clustercentroids); Wst[:, :, 1] = Ws[1]; Wt[:, :, 1] = clustercentroids
			_, clustersilhouettes = silhouette(c, Wt, clusterassignments)
		else
			clusterassignments, clustercentroids, _, clustersilhouettes = cluster_silhouette(c, Hst)  # Exec:623, 637
		end
		ci = clusterassignments[:, 1]
		for (i, ck) in enumerate(ci)                                                  # Exec:631-635
			Wbest[:, i] = WBig[bestIdx][:, ck]
			Hbest[i, :] = HBig[bestIdx][ck, :]
		end
		Wa, Ha, Wv, Hv = cluster_stats(c, Wst, Hst, clusterassignments)               # finalize, Fin:64-77
		Wa = convert(Matrix{T}, Wa); Ha = convert(Matrix{T}, Ha)
		minsilhouette = minimum(clustersilhouettes)                                   # Exec:638
	else
		ifirst = findfirst(idxsol)                                                    # Exec:648 -> Fin:114-118 (unsorted vectors)
		Wa = Statistics.mean(WBig[ifirst]; dims=2); Ha = Statistics.mean(HBig[ifirst]; dims=1)
	end
	if saveall && casefilename != ""                                                  # Exec:650-654
		filename = resultfile(resultdir, casefilename, X, nk, nNMF, "-all")
		mkpath(resultdir)
		JLD.save(filename, "W", WBig, "H", HBig, "Wmean", Wa, "Hmean", Ha, "Wvar", Wv, "Hvar", Hv, "Wbest", Wbest, "Hbest", Hbest, "fit", objvalue, "Cluster Silhouettes", clustersilhouettes, "Cluster assignments", clusterassignments, "Cluster centroids", clustercentroids)
		@info("All results are saved in $(filename)!")
	end
	if best                                                                           # Exec:655-658
		Wa = Wbest; Ha = Hbest
	end
	phi_final = convert(T, frobenius(c, Wa, Ha))                                     # Exec:664-667
	numobservations = length(X) - nancount                                            # Exec:697
	numparameters = length(Wa) + length(Ha)
	aic = 2 * numparameters + numobservations * log(phi_final / numobservations)      # Exec:708
	!quiet && println("Objective function = ", phi_final)
	return Wa, Ha, phi_final, minsilhouette, aic
end

"the restarts of every rank in `ks` on the GPU(s): Dict nk => (WBig, HBig, objvalue, weighted sse)"
function run_restarts(c::Context, X::AbstractMatrix{T}, ks::Vector{Int}, nNMF::Int; weight=1, kw...) where {T <: Number}
	n, m = size(X)
	kwd = Dict{Symbol,Any}(kw)
	p = MuParams()
	for key in MU_KEYS
		haskey(kwd, key) && setproperty!(p, key, kwd[key])
	end
	p.Wfixed = get(kwd, :Wfixed, false) ? 1 : 0
	p.Hfixed = get(kwd, :Hfixed, false) ? 1 : 0
	p.normalize = (haskey(kwd, :Wfixed) || haskey(kwd, :Hfixed)) ? 0 : 1             # modifymatrices, Exec:486-489
	p.weight = weight isa Number ? weight : 1.0
	normalizevector = get(kwd, :normalizevector, Vector{T}(undef, 0))
	Xn = X
	if length(normalizevector) == n                                                   # Mult:27-31 (on a copy: the caller's X stays as it is)
		Xn = X ./ normalizevector
		setX!(c, Xn; lambda=p.lambda)
	end
	setweight!(c, weight, n, m)
	seed = haskey(kwd, :seed) ? kwd[:seed] : nothing                                  # Exec:533 `haskey(kw, :seed)`
	Winit = get(kwd, :Winit, Matrix{Float32}(undef, 0, 0)); Hinit = get(kwd, :Hinit, Matrix{Float32}(undef, 0, 0))
	(sizeof(Winit) > 0 || sizeof(Hinit) > 0) && length(ks) > 1 && error("Winit / Hinit can only be given for a single number of signals")
	Wi = Vector{Array{Float32,3}}(undef, length(ks)); Hi = Vector{Array{Float32,3}}(undef, length(ks))
	for (q, nk) in enumerate(ks)                                                      # reference order: k ascending, restart ascending, W then H
		Wi[q], Hi[q] = draw_inits(n, m, nk, nNMF; seed=seed, Winit=Winit, Hinit=Hinit)
	end
	Wo, Ho, fo, so = mu_sweep(c, n, m, ks, nNMF, p, Wi, Hi)
	out = Dict{Int,Any}()
	for (q, nk) in enumerate(ks)
		WBig = [convert(Matrix{T}, Wo[q][:, :, r]) for r in 1:nNMF]                   # Exec:529-531
		HBig = [convert(Matrix{T}, Ho[q][:, :, r]) for r in 1:nNMF]
		objvalue = convert(Vector{T}, fo[q])
		if length(normalizevector) == n                                               # Mult:119-122, then the objective of Exec:791-792 on X
			for r in 1:nNMF
				WBig[r] .*= normalizevector
			end
		end
		out[nk] = (WBig, HBig, objvalue, so[q])
	end
	if length(normalizevector) == n
		setX!(c, X; lambda=p.lambda)
		for nk in ks, r in 1:nNMF
			out[nk][3][r] = frobenius(c, out[nk][1][r], out[nk][2][r])
		end
	end
	return out
end

"execute_run(X, nk, nNMF; ...) (Exec:483-711) -> (Wa, Ha, phi_final, minsilhouette, aic)"
function execute_run(X::AbstractMatrix{T}, nk::Int, nNMF::Int; clusterWmatrix::Bool=false, acceptratio::Number=1, acceptfactor::Number=Inf,
		quiet::Bool=global_quiet, veryquiet::Bool=true, best::Bool=true, transpose::Bool=false, serial::Bool=false,
		deltas::AbstractMatrix{T}=Matrix{T}(undef, 0, 0), ratios::AbstractMatrix{T}=Matrix{T}(undef, 0, 0), mixture::Symbol=:null,
		resultdir::AbstractString=".", casefilename::AbstractString="", nanaction::Symbol=:zeroed, loadall::Bool=false, saveall::Bool=false,
		weight=1, ngpus::Integer=1, device::Integer=0, context::Union{Nothing,Context}=nothing, kw...) where {T <: Number}
	@assert typeof(weight) <: Number || length(weight) == size(X, 1) || size(weight, 2) == size(X, 2) || size(weight) == size(X)
	(transpose || sizeof(deltas) > 0 || sizeof(ratios) > 0 || mixture != :null) && error("transpose / deltas / ratios / mixture belong to other solvers; use NMFk.execute")
	quiet = veryquiet ? true : quiet
	c = isnothing(context) ? Context(; device=device, ngpus=ngpus) : context
	nancount = isnothing(context) ? setX!(c, X; lambda=get(kw, :lambda, 1e-32)) : count(isnan, X)
	runflag = true
	local WBig, HBig, objvalue, wsse
	if loadall && casefilename != ""                                                  # Exec:499-509
		filename = resultfile(resultdir, casefilename, X, nk, nNMF, "-all")
		if isfile(filename)
			@info("All results are loaded from $(filename)!")
			WBig, HBig, objvalue = JLD.load(filename, "W", "H", "fit")
			wsse = Float64.(objvalue) .^ 2
			saveall = false
			runflag = false
		else
			@warn("File $(filename) with ALL results is missing; runs will be executed!")
		end
	end
	if runflag
		WBig, HBig, objvalue, wsse = run_restarts(c, X, [nk], nNMF; weight=weight, kw...)[nk]
	end
	return execute_run_post(c, X, nk, nNMF, WBig, HBig, objvalue, wsse, nancount; clusterWmatrix=clusterWmatrix, acceptratio=acceptratio,
		acceptfactor=acceptfactor, quiet=quiet, veryquiet=veryquiet, best=best, resultdir=resultdir, casefilename=casefilename,
		nanaction=nanaction, saveall=saveall)
end

"NMFk.execute for a range of k (Exec:178-233) -> (W, H, fitquality, robustness, aic, kopt); `ngpus`: GPUs of this node to use"
function execute(X::AbstractArray{T,N}, nkrange::Union{Vector{Int},AbstractUnitRange{Int}}, nNMF::Integer=10; cutoff::Number=0.5,
		clusterWmatrix::Bool=false, mixture::Symbol=:null, method::Symbol=:simple, algorithm::Symbol=:multdiv, resultdir::AbstractString=".",
		load::Bool=true, save::Bool=true, casefilename::AbstractString="", dims=1:2, ngpus::Integer=1, device::Integer=0, kw...) where {T <: Number, N}
	load, save, casefilename, mixture, method, algorithm, clusterWmatrix = input_checks(X, load, save, casefilename, mixture, method, algorithm, clusterWmatrix)
	.*(size(X)...) == 0 && error("Input array has a zero dimension! Array size=$(size(X))")
	if save                                                                           # Exec:185-192: the matrix next to its results, key "X"
		xfile = joinpath(resultdir, "$(casefilename == "" ? "nmfk" : casefilename)_x_matrix_$(join(size(X), "_")).jld")
		isdir(resultdir) || mkpath(resultdir)
		JLD.save(xfile, "X", X)
	end
	maxk = maximum(collect(nkrange))
	W = Vector{Matrix{T}}(undef, maxk); H = Vector{Matrix{T}}(undef, maxk)
	fitquality = zeros(T, maxk); robustness = zeros(T, maxk); aic = zeros(T, maxk)
	fitquality[1] = Inf; robustness[1] = -1                                           # Exec:200-201
	c = Context(; device=device, ngpus=ngpus)
	nancount = setX!(c, X; lambda=get(kw, :lambda, 1e-32))
	results = execute_many(c, X, collect(nkrange), Int(nNMF), nancount; clusterWmatrix=clusterWmatrix, resultdir=resultdir, load=load, save=save,
		casefilename=casefilename, kw...)
	for nk in nkrange                                                                 # Exec:203-205
		W[nk], H[nk], fitquality[nk], robustness[nk], aic[nk] = results[nk]
	end
	if all(isinf, fitquality[nkrange])                                                # Exec:206-208
		@warn("No successful NMFk runs!")
		return W, H, fitquality, robustness, aic, 0
	end
	@info("Results")
	for nk in nkrange                                                                 # Exec:211-224
		fit = size(W[nk]) == (size(X, 1), nk) && size(H[nk]) == (nk, size(X, 2)) ? frobenius(c, W[nk], H[nk]) : Inf
		abs(fit - fitquality[nk]) > eps(Float16) && @warn("Fit quality is not consistent: $(fit) != $(fitquality[nk])")
		fitquality[nk] = fit
		println("Signals: $(Printf.@sprintf("%2d", nk)) Fit: $(Printf.@sprintf("%12.7g", fitquality[nk])) Silhouette: $(Printf.@sprintf("%12.7g", robustness[nk])) AIC: $(Printf.@sprintf("%12.7g", aic[nk]))")  # Exec:223
	end
	kopt = getk(nkrange, robustness[nkrange], cutoff)                                 # Exec:225
	isnothing(kopt) ? @warn("No optimal solutions") : @info("Optimal solution: $kopt signals")  # Exec:226-230
	return W, H, fitquality, robustness, aic, kopt
end

"NMFk.execute for one k (Exec:236-329) -> (W, H, fitquality, robustness, aic)"
function execute(X::AbstractArray{T,N}, nk::Integer, nNMF::Integer=10; clusterWmatrix::Bool=false, mixture::Symbol=:null, method::Symbol=:simple,
		algorithm::Symbol=:multdiv, resultdir::AbstractString=".", casefilename::AbstractString="", loadonly::Bool=false, load::Bool=true,
		save::Bool=true, quiet::Bool=false, check_inputs::Bool=true, ordersignals::Bool=true, dims=1:2, ngpus::Integer=1, device::Integer=0,
		kw...) where {T <: Number, N}
	if check_inputs
		load, save, casefilename, mixture, method, algorithm, clusterWmatrix = input_checks(X, load, save, casefilename, mixture, method, algorithm, clusterWmatrix)
	end
	.*(size(X)...) == 0 && error("Input array has a zero dimension! Array size=$(size(X))")  # Exec:242-244
	c = Context(; device=device, ngpus=ngpus)
	nancount = setX!(c, X; lambda=get(kw, :lambda, 1e-32))
	return execute_many(c, X, [Int(nk)], Int(nNMF), nancount; clusterWmatrix=clusterWmatrix, resultdir=resultdir, load=load, save=save,
		casefilename=casefilename, loadonly=loadonly, quiet=quiet, ordersignals=ordersignals, kw...)[Int(nk)]
end

"The body of Exec:236-329 for every rank of `ks`, with ONE GPU sweep over all ranks that are not served from the cache."
function execute_many(c::Context, X::AbstractMatrix{T}, ks::Vector{Int}, nNMF::Int, nancount::Int; clusterWmatrix::Bool=false,
		resultdir::AbstractString=".", load::Bool=true, save::Bool=true, casefilename::AbstractString="", loadonly::Bool=false, quiet::Bool=false,
		ordersignals::Bool=true, acceptratio::Number=1, acceptfactor::Number=Inf, veryquiet::Bool=true, best::Bool=true, nanaction::Symbol=:zeroed,
		loadall::Bool=false, saveall::Bool=false, weight=1, kw...) where {T <: Number}
	global first_warning
	if first_warning                                                                  # Mult:8-15
		minimum(sum(X; dims=2)) == 0 && @warn("All matrix entries in a row should not be 0!")
		minimum(sum(X; dims=1)) == 0 && @warn("All matrix entries in a column should not be 0!")
		first_warning = false
	end
	if loadonly                                                                       # Exec:245-251
		load = true; save = false
	end
	if load || save                                                                   # Exec:256-262
		xsize_str = join(size(X), "_")
		check_x_hash!(X, joinpath(resultdir, "$(casefilename)_x_matrix_$(xsize_str).jld"); quiet=quiet)
	end
	(haskey(kw, :Wfixed) || haskey(kw, :Hfixed)) && (ordersignals = false)            # Exec:305-307
	results = Dict{Int,Any}()
	todo = Int[]
	for nk in ks
		runflag = true
		if load                                                                       # Exec:264-303
			filename = resultfile(resultdir, casefilename, X, nk, nNMF)
			if !isfile(filename)
				filename_old = joinpath(resultdir, "$(casefilename)-$(nk)-$(nNMF).jld")
				isfile(filename_old) && (filename = filename_old)
			end
			if isfile(filename)
				Wl, Hl, fitquality, robustness, aic = JLD.load(filename, "W", "H", "fit", "robustness", "aic")
				if size(Wl) == (size(X, 1), nk) && size(Hl) == (nk, size(X, 2))
					fit = frobenius(c, Wl, Hl)
					if abs(fit - fitquality) > eps(Float16)                           # Exec:274-283
						@warn("Fit quality is not consistent: $(fit) != $(fitquality)")
						fitquality = fit
						JLD.save(filename, "W", Wl, "H", Hl, "fit", fitquality, "robustness", robustness, "aic", aic)
					end
					results[nk] = (Wl, Hl, fitquality, robustness, aic)
					runflag = false
				else
					!quiet && @warn("File $(filename) contains inconsistent results; runs will be executed ...")
				end
			end
		end
		if runflag && loadonly                                                        # Exec:291-298
			results[nk] = (Matrix{T}(undef, 0, 0), Matrix{T}(undef, 0, 0), Inf, -1, -Inf)
			runflag = false
		end
		runflag && push!(todo, nk)
	end
	restarts = Dict{Int,Any}()
	if loadall && casefilename != ""                                                  # Exec:499-509, per rank
		for nk in copy(todo)
			filename = resultfile(resultdir, casefilename, X, nk, nNMF, "-all")
			if isfile(filename)
				@info("All results are loaded from $(filename)!")
				Wb, Hb, ov = JLD.load(filename, "W", "H", "fit")
				restarts[nk] = (Wb, Hb, ov, Float64.(ov) .^ 2, false)
			else
				@warn("File $(filename) with ALL results is missing; runs will be executed!")
			end
		end
	end
	run = [nk for nk in todo if !haskey(restarts, nk)]
	if length(run) > 0
		fresh = run_restarts(c, X, run, nNMF; weight=weight, kw...)
		for nk in run
			restarts[nk] = (fresh[nk]..., saveall)
		end
	end
	for nk in todo
		WBig, HBig, objvalue, wsse, sv = restarts[nk]
		Wk, Hk, fitquality, robustness, aic = execute_run_post(c, X, nk, nNMF, WBig, HBig, objvalue, wsse, nancount; clusterWmatrix=clusterWmatrix,
			acceptratio=acceptratio, acceptfactor=acceptfactor, quiet=quiet, veryquiet=veryquiet, best=best, resultdir=resultdir,
			casefilename=casefilename, nanaction=nanaction, saveall=sv)
		so = ordersignals ? signalorder(Wk, Hk) : collect(1:nk)                       # Exec:311-318
		Wk = Wk[:, so]; Hk = Hk[so, :]
		!quiet && println("Signals: $(Printf.@sprintf("%2d", nk)) Fit: $(Printf.@sprintf("%12.7g", fitquality)) Silhouette: $(Printf.@sprintf("%12.7g", robustness)) AIC: $(Printf.@sprintf("%12.7g", aic)) Signal order: $(so)")  # Exec:322
		if save                                                                       # Exec:323-327
			filename = resultfile(resultdir, casefilename, X, nk, nNMF)
			mkpath(resultdir)
			JLD.save(filename, "W", Wk, "H", Hk, "fit", fitquality, "robustness", robustness, "aic", aic)
			!quiet && @info("Results are saved in $(filename)!")
		end
		results[nk] = (Wk, Hk, fitquality, robustness, aic)
	end
	return results
end

end
