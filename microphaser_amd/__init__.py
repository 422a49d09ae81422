"""microphaser_amd - MI355X-native phasing engine behind microphaser's `somatic` path.

Thin ctypes binding of the C ABI in include/microphaser_hip.h (libmicrophaser_hip.so, built
in-tree by microphaser_amd/csrc/Makefile). Python is plumbing only: everything on the hot path
runs in the HIP kernels; the host side (planner, consumer, ingest, CLI) is C++.

There is no CPU fallback: without the built library import fails, without a GPU the kernels'
entry points fail loudly.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# MP_LIB_DIR: another build of the same library (csrc/Makefile SAN=...: the sanitizer builds of the host side, CPU suite only)
LIB_DIR = os.environ.get("MP_LIB_DIR") or os.path.join(_HERE, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libmicrophaser_hip.so")
CLI_PATH = os.path.join(LIB_DIR, "microphaser")

MODE_SOMATIC = 0
MODE_NORMAL = 1  # `microphaser normal` (src/normal_microphasing.rs): fasta + tsv, no normal_fasta
STREAM_FASTA, STREAM_NORMAL_FASTA, STREAM_TSV, STREAM_ALL = 1, 2, 4, 7   # mp_batch_results_select


def build(verbose=False):
    """Compile the HIP library + CLI for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if os.environ.get("MP_LIB_DIR"):
        return   # a sanitizer build is made by its own make invocation (tools/run_sanitized.sh), never implicitly
    r = subprocess.run(cmd, capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libmicrophaser_hip.so failed:\n" + (r.stdout or "") + (r.stderr or ""))


class SynthConfig(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("n_transcripts", ctypes.c_uint32), ("read_len", ctypes.c_uint32),
                ("depth", ctypes.c_double), ("var_spacing", ctypes.c_double),
                ("indel_rate", ctypes.c_double), ("multiallelic_rate", ctypes.c_double), ("softmask_rate", ctypes.c_double),
                ("mate_rate", ctypes.c_double), ("isoform_rate", ctypes.c_double),
                ("gene_streams", ctypes.c_uint32), ("pad_", ctypes.c_uint32), ("gene_keep", ctypes.c_void_p)]


class GeneBatch(ctypes.Structure):
    """include/microphaser_hip.h mp_gene_batch: the phase_gene-level inputs of a batch of genes as struct-of-arrays."""
    _P = ctypes.c_void_p
    _fields_ = [("n_genes", ctypes.c_uint32), ("pad_", ctypes.c_uint32)] + [(n, ctypes.c_void_p) for n in (
        "gene_id", "gene_name", "chrom", "gene_start", "gene_end", "ref_off", "refseq",
        "tx_off", "tx_id", "tx_strand", "exon_off", "exon_start", "exon_end", "exon_frame",
        "read_off", "r_pos", "r_mapq", "r_flag", "r_cigar_off", "cigar", "r_seq_off", "seq", "qual", "r_qname_hash",
        "var_off", "v_pos", "v_kind", "v_alt", "v_len", "v_is_germline", "v_seq_off", "v_seq", "v_prot_change")]


# (field, numpy dtype, which count gives its length) of the plain arrays of a GeneBatch; strings are handled apart
_GB_ARRAYS = [
    ("gene_start", "u8", "g"), ("gene_end", "u8", "g"), ("ref_off", "u8", "g+1"), ("refseq", "u1", "ref"),
    ("tx_off", "u4", "g+1"), ("tx_strand", "u1", "t"), ("exon_off", "u4", "t+1"), ("exon_start", "u8", "e"), ("exon_end", "u8", "e"), ("exon_frame", "u8", "e"),
    ("read_off", "u8", "g+1"), ("r_pos", "i8", "r"), ("r_mapq", "u1", "r"), ("r_flag", "u2", "r"), ("r_cigar_off", "u8", "r+1"), ("cigar", "u4", "cig"),
    ("r_seq_off", "u8", "r+1"), ("seq", "u1", "b"), ("qual", "u1", "b"), ("r_qname_hash", "u8", "r"),
    ("var_off", "u8", "g+1"), ("v_pos", "u8", "v"), ("v_kind", "u1", "v"), ("v_alt", "u1", "v"), ("v_len", "u8", "v"), ("v_is_germline", "u1", "v"),
    ("v_seq_off", "u8", "v+1"), ("v_seq", "u1", "vs"),
]
_GB_STRINGS = [("gene_id", "g"), ("gene_name", "g"), ("chrom", "g"), ("tx_id", "t"), ("v_prot_change", "v")]


def gene_batch_to_python(gb):
    """Copy an mp_gene_batch (ctypes GeneBatch) into host-owned numpy arrays / lists of bytes: dict field -> value."""
    import numpy as np

    def arr(ptr, dtype, n):
        if not n:
            return np.zeros(0, dtype=dtype)
        nbytes = n * np.dtype(dtype).itemsize
        return np.frombuffer((ctypes.c_char * nbytes).from_address(ptr), dtype=dtype).copy()

    g = gb.n_genes
    out = {"n_genes": g}
    counts = {"g": g, "g+1": g + 1}
    for name in ("ref_off", "tx_off", "read_off", "var_off"):
        dt = "u4" if name == "tx_off" else "u8"
        out[name] = arr(getattr(gb, name), dt, g + 1)
    counts["ref"] = int(out["ref_off"][-1]) if g else 0
    counts["t"] = int(out["tx_off"][-1]) if g else 0
    counts["t+1"] = counts["t"] + 1
    counts["r"] = int(out["read_off"][-1]) if g else 0
    counts["r+1"] = counts["r"] + 1
    counts["v"] = int(out["var_off"][-1]) if g else 0
    counts["v+1"] = counts["v"] + 1
    out["exon_off"] = arr(gb.exon_off, "u4", counts["t+1"])
    counts["e"] = int(out["exon_off"][-1])
    out["r_cigar_off"] = arr(gb.r_cigar_off, "u8", counts["r+1"])
    counts["cig"] = int(out["r_cigar_off"][-1])
    out["r_seq_off"] = arr(gb.r_seq_off, "u8", counts["r+1"])
    counts["b"] = int(out["r_seq_off"][-1])
    out["v_seq_off"] = arr(gb.v_seq_off, "u8", counts["v+1"])
    counts["vs"] = int(out["v_seq_off"][-1])
    for name, dt, cnt in _GB_ARRAYS:
        if name not in out:
            out[name] = arr(getattr(gb, name), dt, counts[cnt])
    for name, cnt in _GB_STRINGS:
        n = counts[cnt]
        ptrs = (ctypes.c_char_p * n).from_address(getattr(gb, name)) if n else []
        out[name] = [bytes(x) if x is not None else b"" for x in ptrs]
    return out


def gene_batch_from_python(d):
    """The inverse: a ctypes GeneBatch whose pointers refer to the arrays of `d` (returned too: keep them alive during the call)."""
    import numpy as np
    gb = GeneBatch()
    gb.n_genes = d["n_genes"]
    keep = []
    for name, dt, _cnt in _GB_ARRAYS:
        a = np.ascontiguousarray(d[name], dtype=dt)
        if a.size == 0:
            a = np.zeros(1, dtype=dt)
        keep.append(a)
        setattr(gb, name, a.ctypes.data)
    for name, _cnt in _GB_STRINGS:
        lst = d[name]
        arr = (ctypes.c_char_p * max(1, len(lst)))(*lst)
        keep.append(arr)
        setattr(gb, name, ctypes.cast(arr, ctypes.c_void_p).value)
    return gb, keep


class RunStats(ctypes.Structure):
    _fields_ = [
        ("k1_ms", ctypes.c_double), ("k2_ms", ctypes.c_double), ("k3_ms", ctypes.c_double), ("k3b_ms", ctypes.c_double), ("total_ms", ctypes.c_double),
        ("n_windows_planned", ctypes.c_uint64),
        ("n_steps", ctypes.c_uint64), ("n_transcripts", ctypes.c_uint64), ("n_reads", ctypes.c_uint64), ("n_variants", ctypes.c_uint64),
        ("n_groups", ctypes.c_uint64), ("n_records", ctypes.c_uint64),
        ("bytes_k1", ctypes.c_uint64), ("bytes_k2", ctypes.c_uint64), ("bytes_k3", ctypes.c_uint64), ("bytes_k3b", ctypes.c_uint64),
        ("hbm_bytes", ctypes.c_uint64),
        ("rows_per_lane", ctypes.c_uint32), ("mask_words", ctypes.c_uint32), ("attempts", ctypes.c_uint32), ("pad_", ctypes.c_uint32),
        ("k2seq_ms", ctypes.c_double), ("k2a_ms", ctypes.c_double), ("k2w_ms", ctypes.c_double),
        ("bytes_k2seq", ctypes.c_uint64), ("bytes_k2a", ctypes.c_uint64), ("bytes_k2w", ctypes.c_uint64),
        ("n_steps_seq", ctypes.c_uint64), ("n_steps_w", ctypes.c_uint64), ("n_adm", ctypes.c_uint64),
        ("k2l_ms", ctypes.c_double), ("bytes_k2l", ctypes.c_uint64), ("n_windows_lane", ctypes.c_uint64), ("n_windows_wave", ctypes.c_uint64),
        ("n_groups_k3", ctypes.c_uint64), ("k2win_ms", ctypes.c_double),
        ("n_windows_device", ctypes.c_uint64), ("n_groups_k3a", ctypes.c_uint64), ("n_ids", ctypes.c_uint64),
        ("n_groups_k3c", ctypes.c_uint64), ("n_groups_k3d", ctypes.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    """The loaded C-ABI library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or make -C microphaser_amd/csrc). There is no fallback path." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, cp, u32, u64, i32, dbl = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, ctypes.c_double
    pp = ctypes.POINTER(vp)
    sig = {
        "mp_batch_results_dump": (i32, [vp, vp, ctypes.c_char_p]),
        "mp_batch_results_from_dump": (i32, [vp, vp, ctypes.c_char_p, ctypes.c_uint32, pp]),
        "mp_create": (i32, [i32, pp]),
        "mp_destroy": (None, [vp]),
        "mp_last_error": (cp, [vp]),
        "mp_dataset_load": (i32, [vp, cp, cp, cp, cp, i32, pp]),
        "mp_dataset_synth": (i32, [vp, u64, u32, dbl, dbl, pp]),
        "mp_dataset_synth_ex": (i32, [vp, ctypes.POINTER(SynthConfig), pp]),
        "mp_synth_gene_costs": (i32, [vp, ctypes.POINTER(SynthConfig), ctypes.POINTER(u64)]),
        "mp_dataset_from_arrays": (i32, [vp, ctypes.POINTER(GeneBatch), pp]),
        "mp_dataset_to_arrays": (i32, [vp, vp, i32, ctypes.POINTER(ctypes.POINTER(GeneBatch))]),
        "mp_gene_batch_free": (None, [ctypes.POINTER(GeneBatch)]),
        "mp_dataset_gene_costs": (i32, [vp, vp, ctypes.POINTER(u64)]),
        "mp_batch_create_genes": (i32, [vp, vp, i32, u64, ctypes.POINTER(u32), u32, pp]),
        "mp_results_gene_offsets": (vp, [vp, i32, ctypes.POINTER(ctypes.c_size_t)]),
        "mp_translate": (i32, [vp, cp, cp, u64, u32, vp, vp]),
        "mp_peptides_union": (i32, [vp, ctypes.POINTER(vp), ctypes.POINTER(u64), u32, u32, pp]),
        "mp_dataset_write": (i32, [vp, vp, cp]),
        "mp_dataset_num_genes": (u32, [vp]),
        "mp_dataset_num_reads": (u64, [vp]),
        "mp_dataset_free": (None, [vp]),
        "mp_batch_create": (i32, [vp, vp, i32, u64, u32, u32, pp]),
        "mp_batch_run": (i32, [vp, vp, ctypes.POINTER(RunStats)]),
        "mp_batch_results": (i32, [vp, vp, pp]),
        "mp_batch_results_select": (i32, [vp, vp, u32, pp]),
        "mp_batch_free": (None, [vp]),
        "mp_phase_dataset": (i32, [vp, vp, i32, u64, pp]),
        "mp_results_fasta": (vp, [vp, ctypes.POINTER(ctypes.c_size_t)]),
        "mp_results_normal_fasta": (vp, [vp, ctypes.POINTER(ctypes.c_size_t)]),
        "mp_results_tsv": (vp, [vp, ctypes.POINTER(ctypes.c_size_t)]),
        "mp_results_windows": (u64, [vp]),
        "mp_results_free": (None, [vp]),
        "mp_build_reference": (i32, [vp, cp, u32, pp]),
        "mp_build_reference_buffer": (i32, [vp, cp, ctypes.c_size_t, u32, pp]),
        "mp_peptidome_from_buffer": (i32, [vp, cp, ctypes.c_size_t, u32, pp]),
        "mp_peptides_fasta": (vp, [vp, ctypes.POINTER(ctypes.c_size_t)]),
        "mp_peptides_binary": (vp, [vp, ctypes.POINTER(ctypes.c_size_t)]),
        "mp_peptides_keys": (vp, [vp, ctypes.POINTER(ctypes.c_size_t)]),
        "mp_peptides_count": (u64, [vp]),
        "mp_peptides_free": (None, [vp]),
        "mp_filter": (i32, [vp, cp, cp, u32, pp]),
        "mp_filter_buffers": (i32, [vp, cp, ctypes.c_size_t, cp, ctypes.c_size_t, u32, pp]),
        "mp_filter_peptides": (i32, [vp, cp, ctypes.c_size_t, vp, pp]),
        "mp_filtered_fasta": (vp, [vp, ctypes.POINTER(ctypes.c_size_t)]),
        "mp_filtered_normal_fasta": (vp, [vp, ctypes.POINTER(ctypes.c_size_t)]),
        "mp_filtered_tsv": (vp, [vp, ctypes.POINTER(ctypes.c_size_t)]),
        "mp_filtered_removed_tsv": (vp, [vp, ctypes.POINTER(ctypes.c_size_t)]),
        "mp_filtered_removed_fasta": (vp, [vp, ctypes.POINTER(ctypes.c_size_t)]),
        "mp_filtered_count": (u64, [vp, i32]),
        "mp_filtered_free": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


C_ABI_SYMBOLS = [
    "mp_create", "mp_destroy", "mp_last_error", "mp_dataset_load", "mp_dataset_synth", "mp_dataset_synth_ex", "mp_dataset_write",
    "mp_dataset_num_genes", "mp_dataset_num_reads", "mp_dataset_free", "mp_batch_create", "mp_batch_run",
    "mp_batch_results", "mp_batch_free", "mp_phase_dataset", "mp_results_fasta", "mp_results_normal_fasta",
    "mp_results_tsv", "mp_results_windows", "mp_results_free",
    "mp_build_reference", "mp_peptides_fasta", "mp_peptides_binary", "mp_peptides_keys", "mp_peptides_count", "mp_peptides_free",
    "mp_filter", "mp_filter_buffers", "mp_filter_peptides", "mp_batch_results_select", "mp_filtered_fasta", "mp_filtered_normal_fasta", "mp_filtered_tsv", "mp_filtered_removed_tsv",
    "mp_filtered_removed_fasta", "mp_filtered_count", "mp_filtered_free",
    "mp_synth_gene_costs", "mp_dataset_from_arrays", "mp_dataset_to_arrays", "mp_gene_batch_free", "mp_dataset_gene_costs",
    "mp_batch_create_genes", "mp_results_gene_offsets", "mp_translate", "mp_peptides_union", "mp_build_reference_buffer", "mp_peptidome_from_buffer",
    "mp_batch_results_dump", "mp_batch_results_from_dump",
]


class MicrophaserError(RuntimeError):
    pass


class Context:
    """One per GPU (device >= 0) or host-only (device = -1: planning works, kernels fail loudly)."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        rc = lib().mp_create(device, ctypes.byref(self._h))
        if rc != 0:
            msg = lib().mp_last_error(self._h).decode()
            lib().mp_destroy(self._h)
            self._h = None
            raise MicrophaserError(msg)

    def _check(self, rc):
        if rc != 0:
            raise MicrophaserError(lib().mp_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib().mp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load(self, bam, vcf, fasta, gtf, unsupported_allele_warning_only=False):
        h = ctypes.c_void_p()
        # gtf = None: the annotation is read from stdin, as the reference's sub-commands take it
        self._check(lib().mp_dataset_load(self._h, bam.encode(), vcf.encode(), fasta.encode(), gtf.encode() if gtf is not None else None,
                                          int(unsupported_allele_warning_only), ctypes.byref(h)))
        return Dataset(self, h)

    def peptidome(self, fasta_bytes, peptide_len=9, lazy=True):
        """The peptidome of a nucleotide FASTA (bytes) without the translated FASTA text: Peptides with keys_np (and binary on demand)."""
        h = ctypes.c_void_p()
        self._check(lib().mp_peptidome_from_buffer(self._h, fasta_bytes, len(fasta_bytes), peptide_len, ctypes.byref(h)))
        return Peptides(h, with_binary=not lazy)

    def build_reference(self, fasta_path, peptide_len=9):
        """`microphaser build_reference`: translation + peptide de-duplication on the GPU."""
        h = ctypes.c_void_p()
        if isinstance(fasta_path, bytes):   # the FASTA's bytes instead of a path
            self._check(lib().mp_build_reference_buffer(self._h, fasta_path, len(fasta_path), peptide_len, ctypes.byref(h)))
        else:
            self._check(lib().mp_build_reference(self._h, fasta_path.encode(), peptide_len, ctypes.byref(h)))
        return Peptides(h)

    def filter(self, tsv, reference_binary, peptide_len=9):
        """`microphaser filter`: `tsv` / `reference_binary` are file paths (str) or the files' bytes; `reference_binary` may also be
        the Peptides object of build_reference / peptides_union (filters at that peptidome's peptide length)."""
        h = ctypes.c_void_p()
        if isinstance(reference_binary, Peptides):
            assert isinstance(tsv, bytes) and reference_binary._h
            self._check(lib().mp_filter_peptides(self._h, tsv, len(tsv), reference_binary._h, ctypes.byref(h)))
        elif isinstance(tsv, bytes) and isinstance(reference_binary, bytes):
            self._check(lib().mp_filter_buffers(self._h, tsv, len(tsv), reference_binary, len(reference_binary), peptide_len, ctypes.byref(h)))
        else:
            self._check(lib().mp_filter(self._h, tsv.encode(), reference_binary.encode(), peptide_len, ctypes.byref(h)))
        return Filtered(h)

    def synth(self, seed, n_transcripts, depth=30.0, var_spacing=5.4, indel_rate=0.0, multiallelic_rate=0.0, softmask_rate=0.0, read_len=0, mate_rate=0.0,
              isoform_rate=0.0, gene_streams=False, keep=None):
        """Deterministic synthetic exome. gene_streams: every gene has its own random stream, so that `keep` (an iterable of gene
        ordinals, None = all) materialises a subset of the SAME exome - what a rank of a multi-GPU run does."""
        h = ctypes.c_void_p()
        if indel_rate or multiallelic_rate or softmask_rate or read_len or mate_rate or isoform_rate or gene_streams or keep is not None:
            cfg = SynthConfig(seed, n_transcripts, read_len, depth, var_spacing, indel_rate, multiallelic_rate, softmask_rate, mate_rate, isoform_rate,
                              1 if (gene_streams or keep is not None) else 0, 0, None)
            mask = None
            if keep is not None:
                mask = (ctypes.c_uint8 * n_transcripts)()
                for g in keep:
                    mask[g] = 1
                cfg.gene_keep = ctypes.cast(mask, ctypes.c_void_p)
            self._check(lib().mp_dataset_synth_ex(self._h, ctypes.byref(cfg), ctypes.byref(h)))
        else:
            self._check(lib().mp_dataset_synth(self._h, seed, n_transcripts, depth, var_spacing, ctypes.byref(h)))
        return Dataset(self, h)

    def from_arrays(self, arrays):
        """A data set from decoded records (the phase_gene seam, mp_dataset_from_arrays): `arrays` as gene_batch_to_python returns it."""
        gb, keep = gene_batch_from_python(arrays)
        h = ctypes.c_void_p()
        self._check(lib().mp_dataset_from_arrays(self._h, ctypes.byref(gb), ctypes.byref(h)))
        del keep
        return Dataset(self, h)

    def translate(self, nt, reverse, peptide_len):
        """to_protein on the GPU for len(reverse) windows of 3 * peptide_len nucleotides: (amino acids bytes, list of keys)."""
        n = len(reverse)
        assert len(nt) == n * 3 * peptide_len
        aa = ctypes.create_string_buffer(max(1, n * peptide_len))
        keys = (ctypes.c_uint64 * max(1, n))()
        self._check(lib().mp_translate(self._h, bytes(nt), bytes(bytearray(reverse)), n, peptide_len, ctypes.cast(aa, ctypes.c_void_p), ctypes.cast(keys, ctypes.c_void_p)))
        return aa.raw[:n * peptide_len], list(keys)[:n]

    def peptides_union(self, key_arrays, peptide_len):
        """Union of sorted distinct key arrays (one per rank) -> Peptides (keys, binary)."""
        import numpy as np
        arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in key_arrays]
        ptrs = (ctypes.c_void_p * max(1, len(arrs)))(*[a.ctypes.data if a.size else None for a in arrs])
        counts = (ctypes.c_uint64 * max(1, len(arrs)))(*[a.size for a in arrs])
        h = ctypes.c_void_p()
        self._check(lib().mp_peptides_union(self._h, ptrs, counts, len(arrs), peptide_len, ctypes.byref(h)))
        return Peptides(h, with_binary=False)     # .binary is encoded when somebody asks for it

    def synth_gene_costs(self, seed, n_transcripts, depth=30.0, var_spacing=5.4, read_len=0):
        """Work estimate per gene (CDS nt + one read length per exon, i.e. ~ reads and windows) of the gene_streams exome, without generating it."""
        cfg = SynthConfig(seed, n_transcripts, read_len, depth, var_spacing, 0.0, 0.0, 0.0, 0.0, 0.0, 1, 0, None)
        out = (ctypes.c_uint64 * n_transcripts)()
        self._check(lib().mp_synth_gene_costs(self._h, ctypes.byref(cfg), out))
        return list(out)


def _bytes_at(p, n):
    return bytes((ctypes.c_char * n).from_address(p)) if n else b""


class Peptides:
    """Result of `build_reference`: translated FASTA, the sorted distinct u64 keys, and (on demand) the bincode peptide set.
    The library handle lives as long as the object, so that `Context.filter` can take the peptidome without a bincode round trip."""

    def __init__(self, h, with_binary=True):
        L = lib()
        self._h = h
        n = ctypes.c_size_t()
        p = L.mp_peptides_fasta(h, ctypes.byref(n))
        self.fasta = _bytes_at(p, n.value)
        self._binary = None
        p = L.mp_peptides_keys(h, ctypes.byref(n))
        import numpy as np
        # the keys as a numpy array (a whole-exome peptidome holds ~10^7 of them); `.keys` gives the same as a Python list on demand
        self.keys_np = np.frombuffer((ctypes.c_char * (8 * n.value)).from_address(p), dtype=np.uint64).copy() if n.value else np.zeros(0, dtype=np.uint64)
        self._keys = None
        self.count = L.mp_peptides_count(h)
        if with_binary:
            self.binary

    @property
    def binary(self):
        if self._binary is None:
            n = ctypes.c_size_t()
            p = lib().mp_peptides_binary(self._h, ctypes.byref(n))
            self._binary = _bytes_at(p, n.value)
        return self._binary

    @property
    def keys(self):
        if self._keys is None:
            self._keys = self.keys_np.tolist()
        return self._keys

    def close(self):
        if self._h:
            lib().mp_peptides_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Filtered:
    """Result of `filter`: the five output streams of the reference's sub-command."""

    def __init__(self, h):
        L = lib()
        n = ctypes.c_size_t()
        for name in ("fasta", "normal_fasta", "tsv", "removed_tsv", "removed_fasta"):
            p = getattr(L, "mp_filtered_" + name)(h, ctypes.byref(n))
            setattr(self, name, _bytes_at(p, n.value))
        self.rows, self.peptides, self.groups, self.kept, self.removed = (L.mp_filtered_count(h, k) for k in range(5))
        L.mp_filtered_free(h)


def key_to_peptide(key, length):
    return "".join(chr(65 + ((key >> (5 * (length - 1 - j))) & 31)) for j in range(length))


def keys_to_bincode(keys, length):
    """bincode v1 HashSet<Vec<u8>> (reference: src/peptides.rs:183): u64 count, then u64 length + bytes per peptide."""
    import struct
    out = [struct.pack("<Q", len(keys))]
    for k in keys:
        out.append(struct.pack("<Q", length))
        out.append(key_to_peptide(k, length).encode())
    return b"".join(out)


def decode_bincode_set(data):
    import struct
    n = struct.unpack_from("<Q", data, 0)[0]
    off, out = 8, set()
    for _ in range(n):
        l = struct.unpack_from("<Q", data, off)[0]
        out.add(data[off + 8: off + 8 + l])
        off += 8 + l
    assert off == len(data)
    return out


class Dataset:
    def __init__(self, ctx, h):
        self.ctx, self._h = ctx, h

    @property
    def num_genes(self):
        return lib().mp_dataset_num_genes(self._h)

    @property
    def num_reads(self):
        return lib().mp_dataset_num_reads(self._h)

    def write(self, prefix):
        self.ctx._check(lib().mp_dataset_write(self.ctx._h, self._h, prefix.encode()))

    def batch(self, window_len=27, gene_lo=0, gene_hi=None, mode=MODE_SOMATIC):
        h = ctypes.c_void_p()
        hi = self.num_genes if gene_hi is None else gene_hi
        self.ctx._check(lib().mp_batch_create(self.ctx._h, self._h, mode, window_len, gene_lo, hi, ctypes.byref(h)))
        return Batch(self.ctx, h, self)

    def batch_genes(self, genes, window_len=27, mode=MODE_SOMATIC):
        """Plan + pack an arbitrary ascending list of genes (a shard of a cost-balanced partition)."""
        genes = list(genes)
        arr = (ctypes.c_uint32 * max(1, len(genes)))(*genes)
        h = ctypes.c_void_p()
        self.ctx._check(lib().mp_batch_create_genes(self.ctx._h, self._h, mode, window_len, arr, len(genes), ctypes.byref(h)))
        return Batch(self.ctx, h, self)

    def gene_costs(self):
        """Work estimate per gene (coding nt x read depth), for sharding genes over GPUs."""
        n = self.num_genes
        out = (ctypes.c_uint64 * max(1, n))()
        self.ctx._check(lib().mp_dataset_gene_costs(self.ctx._h, self._h, out))
        return list(out)[:n]

    def to_arrays(self, mode=MODE_SOMATIC):
        """The phase_gene-level inputs of every gene as host-owned arrays (gene_batch_to_python of mp_dataset_to_arrays)."""
        p = ctypes.POINTER(GeneBatch)()
        self.ctx._check(lib().mp_dataset_to_arrays(self.ctx._h, self._h, mode, ctypes.byref(p)))
        try:
            return gene_batch_to_python(p.contents)
        finally:
            lib().mp_gene_batch_free(p)

    def phase(self, window_len=27, mode=MODE_SOMATIC):
        """`microphaser somatic` on this data set: returns Results (fasta, normal_fasta, tsv)."""
        h = ctypes.c_void_p()
        self.ctx._check(lib().mp_phase_dataset(self.ctx._h, self._h, mode, window_len, ctypes.byref(h)))
        return Results(h)

    def close(self):
        if self._h:
            lib().mp_dataset_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    def __init__(self, ctx, h, ds):
        self.ctx, self._h, self._ds = ctx, h, ds

    def run(self):
        st = RunStats()
        self.ctx._check(lib().mp_batch_run(self.ctx._h, self._h, ctypes.byref(st)))
        return st

    def results(self, streams=STREAM_ALL):
        """The output streams; `streams` (STREAM_FASTA | STREAM_NORMAL_FASTA | STREAM_TSV) leaves the ones nobody reads unwritten."""
        h = ctypes.c_void_p()
        self.ctx._check(lib().mp_batch_results_select(self.ctx._h, self._h, streams, ctypes.byref(h)))
        return Results(h)

    def dump_results(self, path):
        """Write the device results of the last run() to a file (the seam between the device pass and the host consumer)."""
        self.ctx._check(lib().mp_batch_results_dump(self.ctx._h, self._h, path.encode()))

    def results_from_dump(self, path, streams=STREAM_ALL):
        """Consume a dump_results() file with this batch (planned from the same inputs): works on a host-only context."""
        h = ctypes.c_void_p()
        self.ctx._check(lib().mp_batch_results_from_dump(self.ctx._h, self._h, path.encode(), streams, ctypes.byref(h)))
        return Results(h)

    def close(self):
        if self._h:
            lib().mp_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Results:
    """Output streams of one run. Buffers can exceed 2 GiB on whole-exome inputs, so they are copied out of the
    library lazily (and only if asked for)."""

    def __init__(self, h):
        self._h = h
        self.windows = lib().mp_results_windows(h)
        self._cache = {}

    def _get(self, name):
        if name not in self._cache:
            n = ctypes.c_size_t()
            p = getattr(lib(), "mp_results_" + name)(self._h, ctypes.byref(n))
            self._cache[name] = bytes((ctypes.c_char * n.value).from_address(p)) if n.value else b""
        return self._cache[name]

    def size(self, name):
        n = ctypes.c_size_t()
        getattr(lib(), "mp_results_" + name)(self._h, ctypes.byref(n))
        return n.value

    def gene_offsets(self, which):
        """Byte offsets of the batch's genes in a stream (0 fasta, 1 normal fasta, 2 tsv): n_genes + 1 values."""
        n = ctypes.c_size_t()
        p = lib().mp_results_gene_offsets(self._h, which, ctypes.byref(n))
        return list((ctypes.c_uint64 * n.value).from_address(p)) if n.value else []

    @property
    def fasta(self):
        return self._get("fasta")

    @property
    def normal_fasta(self):
        return self._get("normal_fasta")

    @property
    def tsv(self):
        return self._get("tsv")

    def close(self):
        if self._h:
            lib().mp_results_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
