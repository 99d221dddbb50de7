# NMFkHIP.jl -- Julia host shim for libnmfk_hip.so (include/nmfk_hip.h).
#
# STATUS: written against the C ABI; NOT exercised in the build container (no `julia` there, SURVEY.md probe table).
# The Python mirror nmfk.jl_amd/execute.py implements the same orchestration and IS tested on the GPU.
#
# Drop-in use:   import NMFkHIP;  W, H, fit, rob, aic, kopt = NMFkHIP.execute(X, 2:5; save=false)
# keeps the signature and 6-tuple of NMFk.execute (src/NMFkExecute.jl:178-233) for method=:simple.
module NMFkHIP

import Random
import Libdl

const libnmfk = get(ENV, "NMFK_HIP_LIB", joinpath(@__DIR__, "..", "nmfk.jl_amd", "libnmfk_hip.so"))

# nmfk_mu_params (include/nmfk_hip.h) == keyword arguments of NMFmultiplicative (src/NMFkMultiplicative.jl:24)
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

check(rc::Integer) = rc == 0 ? nothing :
	(msg = unsafe_string(ccall((:nmfk_last_error, libnmfk), Cstring, ()));
	 rc == 2 ? throw(ErrorException(msg)) : error("libnmfk_hip: $msg (status $rc)"))

mutable struct Context
	h::Ptr{Cvoid}
	function Context(device::Integer=0)
		ENV["GPU_MAX_HW_QUEUES"] = get(ENV, "GPU_MAX_HW_QUEUES", "24") # before the HIP runtime starts
		r = Ref{Ptr{Cvoid}}(C_NULL)
		check(ccall((:nmfk_create, libnmfk), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r))
		c = new(r[])
		finalizer(x -> ccall((:nmfk_destroy, libnmfk), Cint, (Ptr{Cvoid},), x.h), c)
		return c
	end
end

"NMFpreprocessing! (Mult:3-22): uploads X; throws ErrorException(\"All matrix entries must be nonnegative!\")"
function setX!(c::Context, X::AbstractMatrix{<:Real}; lambda=1e-32)
	Xf = convert(Matrix{Float32}, X)
	nan = Ref{Int64}(0); zero = Ref{Int64}(0)
	GC.@preserve Xf check(ccall((:nmfk_set_X, libnmfk), Cint,
		(Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Int64, Cdouble, Ref{Int64}, Ref{Int64}),
		c.h, Xf, size(Xf, 1), size(Xf, 2), stride(Xf, 2), lambda, nan, zero))
	return nan[], zero[]
end

"All restarts of all ranks: replaces Exec:203 x Exec:535-541.  Winit/Hinit are drawn HERE with Julia's RNG in the
reference's order (k ascending, restart ascending, W then H; Mult:38,48) so that Random.seed!(s) reproduces the
reference's starting points."
function mu_sweep(c::Context, n::Int, m::Int, ks::Vector{Int}, nNMF::Int, p::MuParams)
	Wi = [Array{Float32}(undef, n, k, nNMF) for k in ks]
	Hi = [Array{Float32}(undef, k, m, nNMF) for k in ks]
	for (q, k) in enumerate(ks), r in 1:nNMF
		Wi[q][:, :, r] = rand(n, k)
		Hi[q][:, :, r] = rand(k, m)
	end
	Wo = [Array{Float32}(undef, n, k, nNMF) for k in ks]
	Ho = [Array{Float32}(undef, k, m, nNMF) for k in ks]
	fo = [Vector{Float32}(undef, nNMF) for _ in ks]
	so = [Vector{Float64}(undef, nNMF) for _ in ks]
	io = [Vector{Int32}(undef, nNMF) for _ in ks]
	ro = [Vector{Int32}(undef, nNMF) for _ in ks]
	ptrs(v) = [pointer(a) for a in v]
	GC.@preserve Wi Hi Wo Ho fo so io ro begin
		check(ccall((:nmfk_mu_sweep, libnmfk), Cint,
			(Ptr{Cvoid}, Cint, Ptr{Int32}, Cint, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{UInt64}, Ref{MuParams},
			 Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Int32}}, Ptr{Ptr{Int32}}),
			c.h, length(ks), Int32.(ks), nNMF, ptrs(Wi), ptrs(Hi), C_NULL, p,
			ptrs(Wo), ptrs(Ho), ptrs(fo), ptrs(so), ptrs(io), ptrs(ro)))
	end
	return Wo, Ho, fo, io, ro
end

"clustersolutions + silhouettes (Clus:425-517, Fin:36-66); Hs: k x m x nsol, sorted by objective"
function cluster_silhouette(c::Context, Hs::Array{Float32,3})
	k, m, nsol = size(Hs)
	labels = Matrix{Int32}(undef, k, nsol); cent = Matrix{Float32}(undef, k, m)
	psil = Matrix{Float32}(undef, k, nsol); csil = Vector{Float32}(undef, k)
	GC.@preserve Hs check(ccall((:nmfk_cluster_silhouette, libnmfk), Cint,
		(Ptr{Cvoid}, Cint, Cint, Int64, Ptr{Float32}, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
		c.h, k, nsol, m, Hs, labels, cent, psil, csil))
	return labels, cent, psil, csil
end

function frobenius(c::Context, W::Matrix{Float32}, H::Matrix{Float32})
	out = Ref{Float64}(0)
	check(ccall((:nmfk_frobenius, libnmfk), Cint, (Ptr{Cvoid}, Cint, Ptr{Float32}, Ptr{Float32}, Ref{Float64}),
		c.h, size(W, 2), W, H, out))
	return out[]
end

"robustkmeans(X, k, repeats) (src/NMFkCluster.jl:172-246) on the GPU; X: d x n, columns = samples"
function robustkmeans(c::Context, X::Matrix{Float32}, k::Integer, repeats::Integer=1000; maxiter::Integer=1000,
		tol::Number=1e-32, seed::Integer=0, compute_silhouettes_flag::Bool=false)
	d, n = size(X)
	assignments = Vector{Int32}(undef, n); centers = Matrix{Float32}(undef, d, k); costs = Vector{Float32}(undef, n)
	counts = Vector{Int32}(undef, k); totalcost = Ref{Float64}(0)
	best = Ref{Int32}(0); iters = Ref{Int32}(0); nclusters = Ref{Int32}(0)
	sil = compute_silhouettes_flag ? Vector{Float32}(undef, n) : Float32[]
	GC.@preserve X sil check(ccall((:nmfk_robustkmeans, libnmfk), Cint,
		(Ptr{Cvoid}, Cint, Int64, Ptr{Float32}, Cint, Cint, Cint, Cdouble, UInt64, Ptr{Int32}, Ptr{Float32}, Ptr{Float32},
		 Ptr{Int32}, Ref{Float64}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ptr{Float64}, Ptr{Float32}),
		c.h, d, n, X, k, repeats, maxiter, tol, UInt64(seed), assignments, centers, costs, counts, totalcost, best, iters,
		nclusters, C_NULL, compute_silhouettes_flag ? pointer(sil) : C_NULL))
	nclusters[] < k && @warn("Robust k-means analysis could not find $k clusters! Only $(nclusters[]) clusters were found.")
	res = (assignments=Int.(assignments), centers=centers[:, 1:nclusters[]], costs=costs, counts=Int.(counts[1:nclusters[]]),
		totalcost=totalcost[], iterations=Int(iters[]))
	return compute_silhouettes_flag ? (res, sil) : res
end

"getk (src/NMFkPostprocess.jl:7-41)"
function getk(nkrange, robustness, cutoff=0.5)
	all(isnan.(robustness)) && return 0
	kn = findlast(r -> r > cutoff, robustness)
	return isnothing(kn) ? nothing : collect(nkrange)[kn]
end

"NMFk.execute(X, nkrange, nNMF; method=:simple) (Exec:178-233) on the GPU"
function execute(X::AbstractMatrix{T}, nkrange::Union{Vector{Int},AbstractUnitRange{Int}}, nNMF::Integer=10;
		cutoff::Number=0.5, method::Symbol=:simple, save::Bool=false, load::Bool=false, quiet::Bool=false,
		maxiter::Int=10000, tol::Float64=1e-19, device::Integer=0, kw...) where {T <: Number}
	method == :simple || error("Unknown method: $method")
	.*(size(X)...) == 0 && error("Input array has a zero dimension! Array size=$(size(X))")
	c = Context(device)
	nancount, _ = setX!(c, X)
	n, m = size(X); ks = collect(nkrange); maxk = maximum(ks)
	p = MuParams(; maxiter=maxiter, tol=tol, kw...)
	Wo, Ho, fo, _, _ = mu_sweep(c, n, m, ks, Int(nNMF), p)
	W = Vector{Matrix{T}}(undef, maxk); H = Vector{Matrix{T}}(undef, maxk)
	fitquality = zeros(T, maxk); robustness = zeros(T, maxk); aic = zeros(T, maxk)
	fitquality[1] = Inf; robustness[1] = -1
	for (q, k) in enumerate(ks)
		idxsort = sortperm(fo[q])                                     # Exec:545
		Wb = Wo[q][:, :, idxsort[1]]; Hb = Ho[q][:, :, idxsort[1]]
		sil = 1.0
		if k > 1
			labels, _, _, csil = cluster_silhouette(c, Ho[q][:, :, idxsort]) # Exec:623, 637
			ci = labels[:, 1]; Wb = Wb[:, ci]; Hb = Hb[ci, :]                  # Exec:631-635
			sil = minimum(csil)                                               # Exec:638
		end
		phi = frobenius(c, Wb, Hb)                                   # Exec:664-667
		nobs = length(X) - nancount
		so = sortperm(vec(sum(Wb; dims=1)) .* vec(sum(Hb; dims=2)); rev=true) # Post:148-158
		W[k] = Wb[:, so]; H[k] = Hb[so, :]
		fitquality[k] = phi; robustness[k] = sil
		aic[k] = 2 * (length(Wb) + length(Hb)) + nobs * log(phi / nobs)    # Exec:697-708
		!quiet && println("Signals: $k Fit: $phi Silhouette: $sil AIC: $(aic[k])")
	end
	kopt = getk(ks, robustness[ks], cutoff)                          # Exec:225
	return W, H, fitquality, robustness, aic, kopt
end

end
