"""
`gretel` command line on top of the MI355X hot path: same positional arguments, options,
stdout table and output files as the reference driver (gretel/cmd.py:11-240), with the
SPINS loop (cmd.py:148-179) executed on the device by `Hansel.spin`.

Files written (formats: docs/protocol.rst:48-77):
    <out>/out.fasta       one record per distinct haplotype, in order of discovery   cmd.py:183-216
    <out>/snp.fasta       the SNP alleles only                                       cmd.py:218-219
    <out>/gretel.crumbs   "# N n_crumbs n_slices L" + one line per haplotype         cmd.py:224-240
"""
from __future__ import annotations

import argparse
import sys

from . import __version__
from . import util
from .hansel import Hansel

MIN_REMOVE = 0.01          # cmd.py:157


def build_parser():
    p = argparse.ArgumentParser(prog="gretel", description="Gretel: A metagenomic haplotyper (MI355X hot path).")
    p.add_argument("bam")
    p.add_argument("vcf")
    p.add_argument("contig")
    p.add_argument("-s", "--start", type=int, default=1, help="1-indexed included start base position [default: 1]")
    p.add_argument("-e", "--end", type=int, default=-1, help="1-indexed included end base position [default: reference length]")
    p.add_argument("-p", "--paths", type=int, default=100, help="maximum number of paths to generate [default: 100]")
    p.add_argument("--master", default=None, help="master FASTA used to fill the non-SNP positions (otherwise --gapchar)")
    p.add_argument("--gapchar", default="N", help="character for non-SNP positions without --master [default: N]")
    p.add_argument("--delchar", default="", help="character written for a deletion [default: nothing]")
    p.add_argument("--quiet", default=False, action="store_true", help="do not print the per-SNP table")
    p.add_argument("-o", "--out", default=".", help="output directory [default: .]")
    p.add_argument("-@", "--threads", type=int, default=1, help="accepted for compatibility (the fill runs on the GPU)")
    p.add_argument("--debugreads", type=str, default="", help="A newline delimited list of read names to output debug data when parsing the BAM")
    p.add_argument("--debugpos", type=str, default="", help="A newline delimited list of 1-indexed genomic positions to output debug data when parsing the BAM")
    p.add_argument("--max-depth", type=int, default=8000, help="read-buffer cap of the pileup: reads beyond it at a position are dropped, as "
                   "the pysam pileup the reference runs on does at its default of 8000 (0 = keep every read) [default: 8000]")
    p.add_argument("--debughpos", type=str, default=",", help="comma delimited 1-indexed SNP ranks to print branch weights for")
    p.add_argument("--dumpmatrix", type=str, default=None, help="dump the Hansel tensor (.npz) to this path")
    p.add_argument("--dumpsnps", type=str, default=None, help="dump the SNP positions to this path")
    p.add_argument("--pepper", action="store_true", help="permissive read filter (pysam stepper 'all' in the reference)")
    p.add_argument("--version", action="version", version="%(prog)s " + __version__)
    return p


def read_first_fasta_record(path):
    """What cmd.py:187-188 takes from pysam.FastaFile: the sequence of the first record."""
    seq = []
    seen = False
    with open(path) as fh:
        for line in fh:
            if line.startswith(">"):
                if seen:
                    break
                seen = True
                continue
            if seen:
                seq.append(line.strip())
    return "".join(seq)


FAIL_TEXT = '''[FAIL] Unable to recover pairwise evidence concerning SNP #%d at position %d
       Gretel needs every SNP to appear on a read with at least one other SNP, at least once.
       There is no read in your data set that bridges SNP #%d with any of its neighbours.

       * If you are trying to run Gretel along an entire contig or genome, please note that
       this is not the recommended usage for Gretel, as it was intended to uncover the
       variation in a metahaplome: the set of haplotypes for a specific gene.
           See our pre-print https://doi.org/10.1101/223404 for more information

       Consider running a prediction tool such as `prokka` on your assembly or reference
       and using the CDS regions in the GFF for corresponding genes of interest to
       uncover haplotypes with Gretel instead.

       * If you are already doing this, consider calling for SNPs more aggressively.
       We use `snpper` (https://github.com/SamStudio8/gretel-test/blob/master/snpper.py)
       to determine any site in a BAM that has at least one read in disagreement with
       the reference as a SNP. Although this introduces noise from alignment and sequence
       error, Gretel is fairly robust. Importantly, this naive calling method will
       likely close gaps between SNPs and permit recovery.

       * Finally, consider that the gaps are indicative that your reads do not support
       one or more parts of your assembly or reference. You could try and find or construct
       a more suitable reference, or reduce the size of the recovery window.

       Sorry :(\n'''            # the message of cmd.py:93-117, character for character

HOLE_TEXT = '''[NOTE] Unable to select next branch from SNP %d to %d
       By design, Gretel will attempt to recover haplotypes until a hole in the graph has been found.
       Recovery will intentionally terminate now.\n'''      # gretel.py:177-179


def gap_report(hansel, vcf_h, out=None):
    """cmd.py:85-118: returns True (and explains, in the reference's words) when some SNP has no pairwise evidence."""
    out = out or sys.stderr
    i = hansel.gap_check()
    if i < 0:
        return False
    pos = vcf_h["snp_rev"][i - 1] if i > 0 else 0
    out.write(FAIL_TEXT % (i, pos, i))
    return True


def print_snp_table(hansel, vcf_h, out=None):
    """cmd.py:123-145"""
    out = out or sys.stdout
    out.write("i\tpos\tgap\tA\tC\tG\tT\tN\t-\t_\ttot\n")
    last = 0
    for i in range(0, vcf_h["N"] + 1):
        c = hansel.counts_array(i)
        pos = vcf_h["snp_rev"][i - 1] if i > 0 else 0
        out.write("%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n" % (
            i, pos, pos - last, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]))
        last = pos


def _add_path(paths, i, key, hansel_path, hp_current, hp_original, magnitude):
    """cmd.py:164-179"""
    if key not in paths:
        paths[key] = {"hp_current": [], "hp_original": [], "i": [], "i_0": i, "n": 0, "magnitude": 0,
                      "hansel_path": hansel_path}
    rec = paths[key]
    rec["n"] += 1
    rec["i"].append(i)
    rec["magnitude"] += magnitude
    rec["hp_current"].append(hp_current)
    rec["hp_original"].append(hp_original)


def recover(hansel, n_snps, max_paths, log=None):
    """cmd.py:148-179 on the device; returns the PATHS table in order of discovery.  The spins have already
    happened when the notes are written, but the notes are the reference's, in the reference's order."""
    log = log or sys.stderr
    res = hansel.spin(max_paths, MIN_REMOVE)
    paths = {}
    for i in range(res["n"]):
        log.write("[NOTE] *Establishing next path\n")                                   # gretel.py:142
        if res["min_marginal"][i] < MIN_REMOVE:                                          # cmd.py:158-160
            log.write("[RWGT] Ratio %.10f too small, adjusting to %.3f\n" % (res["min_marginal"][i], MIN_REMOVE))
        log.write("[RWGT] Ratio %.3f, Removed %.1f\n" % (res["ratio"][i], res["magnitude"][i]))   # gretel.py:97
        _add_path(paths, i, Hansel.path_str(res["paths"][i]), hansel.path_symbols(res["paths"][i]),
                  float(res["hp_current"][i]), float(res["hp_original"][i]), float(res["magnitude"][i]))
    if res["hole_at"]:
        log.write("[NOTE] *Establishing next path\n")
        log.write(HOLE_TEXT % (res["hole_at"] - 1, res["hole_at"]))
    return paths


def recover_with_debug(hansel, n_snps, max_paths, debug_hpos, log=None):
    """The same loop one path at a time through gretel_amd.gretel (one kernel sequence and one host round trip per
    path): what --debughpos needs, because the reference prints the branch weights of EVERY path as it walks
    (gretel.py:147-164), each against the tensor as reweighted so far."""
    from . import gretel
    log = log or sys.stderr
    paths = {}
    for i in range(max_paths):
        init_path, init_prob, init_min = gretel.generate_path(n_snps, hansel, hansel, debug_hpos=debug_hpos)
        if init_path is None:
            break
        if init_min < MIN_REMOVE:
            log.write("[RWGT] Ratio %.10f too small, adjusting to %.3f\n" % (init_min, MIN_REMOVE))
            init_min = MIN_REMOVE
        mag = gretel.reweight_hansel_from_path(hansel, init_path, init_min)
        _add_path(paths, i, "".join(str(x) for x in init_path), init_path, init_prob["hp_current"], init_prob["hp_original"], mag)
    return paths


def write_outputs(paths, hansel, vcf_h, args):
    """cmd.py:181-240, byte for byte."""
    dirn = args.out + "/"
    if args.master:
        master_seq = read_first_fasta_record(args.master)
    else:
        master_seq = [' '] * args.end
    with open(dirn + "out.fasta", "w") as fasta, open(dirn + "snp.fasta", "w") as hfasta:
        for key in sorted(paths, key=lambda x: paths[x]["i_0"]):
            p = paths[key]
            seq = list(master_seq[:])
            for j, mallele in enumerate(p["hansel_path"][1:]):
                pos = vcf_h["snp_rev"][j]
                seq[pos - 1] = args.delchar if mallele == hansel.symbols_d["-"] else mallele
            text = "".join(str(x) for x in seq[args.start - 1: args.end])
            if not args.master:
                text = text.replace(' ', args.gapchar)
            fasta.write(">%d__%.2f\n" % (p["i_0"], p["hp_current"][0]))
            fasta.write("%s\n" % text)
            hfasta.write(">%d__%.2f\n" % (p["i_0"], p["hp_current"][0]))
            hfasta.write("%s\n" % "".join(str(x) for x in p["hansel_path"][1:]))
    with open(dirn + "gretel.crumbs", "w") as crumbs:
        crumbs.write("# %d\t%d\t%d\t%.2f\n" % (vcf_h["N"], hansel.n_crumbs, hansel.n_slices, hansel.L))
        for key in sorted(paths, key=lambda x: paths[x]["hp_current"][0], reverse=True):
            p = paths[key]
            crumbs.write("%d\t%d\t%s\t%s\t%.2f\n" % (
                p["i_0"], p["n"], ",".join("%.2f" % x for x in p["hp_current"]),
                ",".join("%.2f" % x for x in p["hp_original"]), p["magnitude"]))


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.end == -1:
        args.end = util.get_ref_len_from_bam(args.bam, args.contig)               # cmd.py:53-55
        sys.stderr.write("[NOTE] Setting end_pos to %d" % args.end)
    if not (args.debugreads or args.debugpos):
        # (the BAM's blocks are read and inflated on the decoder's threads while the VCF is parsed here: include/gretel_io.h, gio_prefetch)
        util.prefetch_bam(args.bam, args.contig, args.start, args.end)
    vcf_h = util.process_vcf(args.vcf, args.contig, args.start, args.end)         # cmd.py:69
    if args.dumpsnps:                                                             # cmd.py:70-74
        with open(args.dumpsnps, "w") as fh:
            for k in sorted(vcf_h["snp_fwd"].keys()):
                fh.write("%d\t%d\t%d\n" % (vcf_h["snp_fwd"][k] + 1, k, k - args.start + 1))
    debug_reads, debug_pos = set(), set()                                          # cmd.py:57-67
    if args.debugreads:
        with open(args.debugreads) as fh:
            debug_reads = {line.strip() for line in fh}
    if args.debugpos:
        with open(args.debugpos) as fh:
            debug_pos = {int(line.strip()) for line in fh if line.strip()}
    hansel = util.load_from_bam(args.bam, args.contig, args.start, args.end, vcf_h, n_threads=args.threads,
                                debug_reads=debug_reads, debug_pos=debug_pos, max_depth=args.max_depth,
                                stepper="all" if args.pepper else "samtools")     # cmd.py:78
    hansel.snapshot_original()                                                    # cmd.py:79 (what the copy is used for)
    if args.dumpmatrix:
        hansel.save_hansel_dump(args.dumpmatrix)                                  # cmd.py:81-82
    if gap_report(hansel, vcf_h):
        sys.exit(1)                                                               # cmd.py:118
    if not args.quiet:
        print_snp_table(hansel, vcf_h)
    debug_hpos = []
    for x in (args.debughpos or "").split(","):
        try:
            debug_hpos.append(int(x))
        except ValueError:
            pass
    if debug_hpos:
        paths = recover_with_debug(hansel, vcf_h["N"], args.paths, debug_hpos)
    else:
        paths = recover(hansel, vcf_h["N"], args.paths)
    write_outputs(paths, hansel, vcf_h, args)
    try:                        # the decoder's kept working buffers (include/gretel_io.h: GIO_KEEP_MB): a run has one decode
        from . import bamio
        if getattr(bamio, "_io", None) is not None:
            bamio.native_release_buffers()
    except Exception:
        pass
    return 0


if __name__ == "__main__":
    sys.exit(main())
