/*
 * sections.h -- host side of zsc_compress with source_len > max_block_len at levels 1-9.
 *
 * The reference's wrapper (src/zsc_compress.c:121-138) hands deflate() the input in sections of
 * max_block_len with Z_FULL_FLUSH, and the OUTPUT in slices of max_block_len too, refilling
 * whichever has run out before each call.  Normally every section ends with the empty stored
 * block 00 00 FF FF and the history is forgotten (src/deflate.c:1240-1252): the sections are
 * independent deflate runs, which is what the GPU likes -- they are all parsed at once.
 *
 * But deflate() returns whenever the output slice is full, and the wrapper then refills the
 * input as well if all of it has been read into the window.  So the next section is let in
 * early -- the parse carries on with the history and no marker is written -- when the slice
 * runs out while
 *   (0) a block that was cut for being full is flushed, the rest of the section being in the
 *       window already, or
 *   (1) the section's last block is flushed (SURVEY finding 2).
 * Where that happens depends on the compressed sizes of everything before.  A "run" is a
 * maximal sequence of sections parsed with one history; a "joint" (ZdSched) is a place inside
 * a run where the next section was let in.
 *
 * The work goes in rounds.  Round 0 parses every section as the start of a run.  Then SecSim
 * follows the calls of the wrapper and of deflate() for each stream, with the block sizes the
 * GPU found -- pure arithmetic, the counterpart of StoreSim for level 0 -- until it meets a
 * joint the run was not parsed with.  That run is parsed again in the next round, with the
 * joint (and the section it lets in); the other runs stand.
 *
 * A stream needs a round per joint that way, and a parse has a long latency on a GPU.  So
 * runs are parsed with joints of kind 1 they may not have ("speculated"): what is parsed
 * before the end of a section does not depend on whether the run goes on behind it, so a
 * run that turns out to end earlier than it was parsed is simply used up to there (its last
 * byte masked: the bits behind belong to blocks that are dropped).  Round 0 gives every
 * section one such joint when sections are short (most runs there are two sections long); a run
 * that is parsed again gets as many as it has confirmed ones, so an incompressible stream,
 * where every section end is a joint, takes log2(sections) rounds instead of one per section.
 *
 * Joints of kind 0 do not sit at section ends, so they cannot be listed ahead.  But a joint
 * whose cut lies MIN_LOOKAHEAD or more before the old end of the input cannot have been noticed
 * by the parse before it (the end only shows in lookahead caps and fill_window calls, both
 * within MIN_LOOKAHEAD of it): every parser "folds" such a joint -- takes the new length from
 * the start of the phase -- and the host may guess one (ZD_JOINT_ANYWHERE: "this phase goes on
 * for another section").  The simulation then checks every block of a guessed phase: one cut
 * within MIN_LOOKAHEAD of the real end, or the real end reached with the guess still open, and
 * the run is parsed again without it.  With that an incompressible stream, where nearly every
 * output slice ends in a joint, costs a few extra parses instead of one per joint.
 *
 * What fill_window has read (the wrapper's "avail_in == 0") is followed statefully: it is only
 * called when the lookahead runs low, so after a joint of kind 0 the old figure stands until
 * the parse comes within MIN_LOOKAHEAD of the OLD end -- a second block that fills up before
 * that is not a joint.  Block records carry the window end and the position of the parse-loop
 * iteration that cut them (the window only moves inside fill_window).
 *
 * Plain C++, no HIP: the kernels' lane-emulation build (tests/emu) runs the same code.
 */
#ifndef ZSC_SECTIONS_H
#define ZSC_SECTIONS_H

#include <stdint.h>

#include <algorithm>
#include <map>
#include <vector>

#include "zsc_dev.h"

/* a block of a parsed run, as the simulation needs it */
struct SecBlock {
    uint32_t upto;     /* first input position after the block (relative to the run) */
    uint32_t end_bit;  /* first bit after the block in the run's own bit stream */
    uint32_t wend;     /* ZdBlockRec.wend */
    uint32_t at;       /* ZdBlockRec.at */
    uint32_t cut;      /* ZD_CUT_* */
    uint32_t last;
};

struct SecRun {
    uint32_t stream = 0;
    uint32_t start = 0; /* first byte of the run in the stream's input */
    uint32_t n0 = 0;    /* length of its first section */
    uint32_t n = 0;     /* its whole length, all known joints applied */
    bool more = false;  /* the stream goes on after it */
    std::vector<ZdSched> sched;
    uint32_t confirmed = 0; /* joints [0, confirmed) were met by the simulation, the rest are speculated */
    uint32_t no_guess_at = 0xffffffffu; /* a guess of kind 0 failed for the phase that starts with this joint */
    /* filled in by whoever runs the kernels: */
    std::vector<SecBlock> blocks;
    uint32_t round = 0, job = 0; /* where its compressed bytes are */
};

/* the finished stream is put together from these */
enum { SEC_PIECE_HEADER = 0, SEC_PIECE_RUN = 1, SEC_PIECE_MARKER = 2, SEC_PIECE_TRAILER = 3, SEC_PIECE_TAIL = 4 };
struct SecPiece {
    uint32_t kind;
    uint32_t round, job; /* SEC_PIECE_RUN: the first `len` bytes of that job's output;
                            SEC_PIECE_TAIL: byte `src` of it, AND `mask`, then len - 1 zero bytes */
    uint32_t dst, len;   /* where in the stream */
    uint32_t src, mask;
};

struct SecStream {
    uint32_t source_len = 0, max_block_len = 0, dest_cap = 0;
    int wrap = 1;
    uint32_t hdr_len = 0; /* bytes of the zlib / gzip header (a caller's gz_header can be longer than 10) */
    uint32_t need = ZD_MIN_LOOKAHEAD; /* the parse function calls fill_window when fewer bytes of lookahead are
                                         left: MIN_LOOKAHEAD, MAX_MATCH + 1 for Z_RLE, 1 for Z_HUFFMAN_ONLY */
    std::map<uint32_t, SecRun> runs; /* by start */
    /* outcome */
    bool done = false;
    int status = 0;
    int broken_line = 0;
    uint32_t delivered = 0, produced = 0;
    std::vector<SecPiece> pieces;
};

enum { SEC_Z_OK = 0, SEC_Z_STREAM_END = 1, SEC_Z_STREAM_ERROR = -2, SEC_Z_BUF_ERROR = -5 };

/* Follows zsc_compress2's loop (reference src/zsc_compress.c:121-138) and deflate()
 * (src/deflate.c:964-1295) for one stream. */
struct SecSim {
    SecStream &s;
    uint32_t produced = 0, delivered = 0, avail_out = 0;
    uint32_t given = 0;    /* input handed to deflate() so far */
    uint32_t run_abs = 0;  /* where the current run starts */
    uint32_t data_end = 0; /* how far fill_window has read, relative to the run */
    SecRun *run = nullptr;
    uint32_t bi = 0, si = 0;  /* next block / joint of the run */
    uint32_t run_out0 = 0;    /* where the run's bytes start in the stream */
    uint32_t last_end_bit = 0, last_upto = 0, last_cut = 0;
    bool header_done = false, finishing = false, trailer_done = false;
    bool broken = false;
    int broken_line = 0; /* which consistency check failed (diagnostics) */
    std::vector<SecPiece> pieces;

    explicit SecSim(SecStream &stream) : s(stream) {}

    void flush_pending() /* src/deflate.c:840-861; _tr_flush_bits first: whole bytes only */
    {
        const uint32_t have = produced - delivered, len = std::min(have, avail_out);
        delivered += len;
        avail_out -= len;
    }
    void piece(uint32_t kind, uint32_t dst, uint32_t len)
    {
        SecPiece pc;
        pc.kind = kind;
        pc.round = run ? run->round : 0;
        pc.job = run ? run->job : 0;
        pc.dst = dst;
        pc.len = len;
        pc.src = pc.mask = 0;
        if (len)
            pieces.push_back(pc);
    }
    /* more joints than the simulation has met.  kind 1: one per section end.  kind 0: somewhere
     * in the section, far enough from its end for the parsers to fold it (ZD_JOINT_ANYWHERE) --
     * the phase simply goes on for `count` more sections. */
    void speculate(SecRun &r, uint32_t count, uint32_t kind = 1u)
    {
        while (count-- != 0 && r.start + r.n < s.source_len) {
            ZdSched j;
            j.pos = kind == 1u ? r.n : ZD_JOINT_ANYWHERE;
            j.new_n = r.n + std::min(s.max_block_len, s.source_len - (r.start + r.n));
            j.kind = kind;
            j.pad = 0;
            r.sched.push_back(j);
            r.n = j.new_n;
        }
        r.more = r.start + r.n < s.source_len;
    }
    /* was the current phase parsed with more input than deflate() has been given (a guessed
     * joint of kind 0 that has not happened yet)? */
    bool phase_guessed() const
    {
        return run && si < run->sched.size() && run->sched[si].kind == 0u &&
               run->sched[si].pos == ZD_JOINT_ANYWHERE;
    }
    /* the guess did not come true: parse the run again with what is known, no guess for this phase */
    SecRun *retract = nullptr;
    void give_up_guess()
    {
        run->sched.resize(si);
        run->confirmed = std::min<uint32_t>(run->confirmed, si);
        run->n = given - run_abs;
        run->more = given < s.source_len;
        run->no_guess_at = si;
        retract = run;
    }
    void take_block(const SecBlock &b)
    {
        produced = run_out0 + (b.end_bit >> 3);
        last_end_bit = b.end_bit;
        last_upto = b.upto;
        last_cut = b.cut;
        /* how far has fill_window read?  It is only called when the lookahead runs low, so after
         * the input has grown at a joint of kind 0 the old figure stands until the parse comes that
         * close to the OLD end; and the window only moves in fill_window, so its end as recorded
         * at this cut is its end at the last call. */
        if ((uint64_t)data_end < (uint64_t)b.at + s.need)
            data_end = std::min(b.wend, given - run_abs);
        bi++;
    }

    /* one deflate() call */
    int deflate(bool finish)
    {
        if (avail_out == 0)
            return SEC_Z_BUF_ERROR; /* :987-990 */
        if (produced != delivered) { /* :996-1008 */
            flush_pending();
            if (avail_out == 0)
                return SEC_Z_OK;
        }
        if (!header_done) { /* :1029-1090 */
            header_done = true;
            if (s.wrap) {
                piece(SEC_PIECE_HEADER, produced, s.hdr_len);
                produced += s.hdr_len;
                flush_pending();
                if (produced != delivered)
                    return SEC_Z_OK;
            }
        }
        if (!finishing) { /* :1211-1260 */
            if (!run) {
                std::map<uint32_t, SecRun>::iterator it = s.runs.find(run_abs);
                if (it == s.runs.end()) {
                    broken = true, broken_line = __LINE__;
                    return SEC_Z_STREAM_ERROR;
                }
                run = &it->second;
                bi = si = 0;
                run_out0 = produced;
                last_end_bit = 0;
            }
            const uint32_t n_cur = given - run_abs;
            for (;;) {
                if (bi < run->blocks.size() && run->blocks[bi].cut == ZD_CUT_FULL &&
                    run->blocks[bi].upto <= n_cur) {
                    if (phase_guessed() && (uint64_t)run->blocks[bi].upto + ZD_MIN_LOOKAHEAD > n_cur) {
                        give_up_guess(); /* cut too close to the real end to be what the reference does */
                        return SEC_Z_STREAM_ERROR;
                    }
                    take_block(run->blocks[bi]); /* FLUSH_BLOCK(s, 0) inside deflate_slow/_fast/... */
                    flush_pending();
                    if (avail_out == 0)
                        return SEC_Z_OK; /* need_more */
                    continue;
                }
                /* the input given so far is used up */
                if (phase_guessed()) {
                    give_up_guess(); /* the parse went on as if there were more */
                    return SEC_Z_STREAM_ERROR;
                }
                data_end = n_cur;
                if (finish) { /* FLUSH_BLOCK(s, 1) */
                    if (bi >= run->blocks.size() || !run->blocks[bi].last || run->blocks[bi].upto != n_cur) {
                        broken = true, broken_line = __LINE__;
                        return SEC_Z_STREAM_ERROR;
                    }
                    take_block(run->blocks[bi]);
                    produced = run_out0 + ((last_end_bit + 7u) >> 3); /* bi_windup */
                    piece(SEC_PIECE_RUN, run_out0, produced - run_out0);
                    flush_pending();
                    finishing = true;
                    if (avail_out == 0)
                        return SEC_Z_OK; /* finish_started */
                    break;
                }
                if (bi < run->blocks.size() && run->blocks[bi].cut == ZD_CUT_END &&
                    run->blocks[bi].upto == n_cur) {
                    take_block(run->blocks[bi]); /* if (s->last_lit) FLUSH_BLOCK(s, 0) */
                    flush_pending();
                    if (avail_out == 0)
                        return SEC_Z_OK; /* need_more: the marker is never written (finding 2) */
                }
                if ((bi != run->blocks.size() || n_cur != run->n) && si < run->confirmed) {
                    broken = true, broken_line = __LINE__; /* the run was parsed for other joints than the ones met */
                    return SEC_Z_STREAM_ERROR;
                }
                /* block_done with Z_FULL_FLUSH: _tr_stored_block(s, 0, 0, 0), history forgotten.
                 * The run may have been parsed further (speculated joints): its last byte can
                 * hold bits of a block that is not used */
                const uint32_t whole = last_end_bit >> 3, len = (last_end_bit + 3u + 7u) >> 3;
                piece(SEC_PIECE_RUN, run_out0, whole);
                piece(SEC_PIECE_TAIL, run_out0 + whole, len - whole);
                pieces.back().src = whole;
                pieces.back().mask = (1u << (last_end_bit & 7u)) - 1u;
                piece(SEC_PIECE_MARKER, run_out0 + len, 4u);
                produced = run_out0 + len + 4u;
                run_abs += n_cur;
                run = nullptr;
                data_end = 0;
                flush_pending();
                return SEC_Z_OK; /* whether or not avail_out is 0 (:1253-1257, :1262-1264) */
            }
        }
        if (!finish)
            return SEC_Z_OK;
        if (s.wrap == 0)
            return SEC_Z_STREAM_END;
        if (!trailer_done) { /* :1270-1290 */
            trailer_done = true;
            run = nullptr;
            piece(SEC_PIECE_TRAILER, produced, s.wrap == 1 ? 4u : 8u);
            produced += s.wrap == 1 ? 4u : 8u;
            flush_pending();
            return produced != delivered ? SEC_Z_OK : SEC_Z_STREAM_END;
        }
        return SEC_Z_STREAM_END;
    }

    /* the wrapper's loop; returns the run that has to be parsed again with one more joint, or
     * nullptr when the stream is settled (s.status, s.delivered, s.pieces) */
    SecRun *go()
    {
        uint32_t left_dest = s.dest_cap, left_src = s.source_len;
        int err = SEC_Z_OK;
        while (err == SEC_Z_OK) {
            if (avail_out == 0) {
                avail_out = std::min(left_dest, s.max_block_len);
                left_dest -= avail_out;
            }
            if (run_abs + data_end == given) { /* avail_in == 0 */
                const uint32_t take = std::min(left_src, s.max_block_len);
                given += take;
                left_src -= take;
                if (take && run) {
                    /* deflate() came back in the middle of a run with all its input read: a joint */
                    ZdSched j;
                    j.pos = last_upto;
                    j.new_n = given - run_abs;
                    j.kind = last_cut == ZD_CUT_END ? 1u : 0u;
                    j.pad = 0;
                    bool known = si < run->sched.size() && run->sched[si].pos == j.pos &&
                                 run->sched[si].new_n == j.new_n && run->sched[si].kind == j.kind;
                    if (!known && phase_guessed() && j.kind == 0u && run->sched[si].new_n == j.new_n &&
                        (uint64_t)j.pos + ZD_MIN_LOOKAHEAD <= j.new_n - take) {
                        run->sched[si].pos = j.pos; /* the guess came true, here */
                        known = true;
                    }
                    if (known) {
                        si++;
                        run->confirmed = std::max(run->confirmed, si);
                    } else if (si < run->confirmed) {
                        broken = true, broken_line = __LINE__;
                        err = SEC_Z_STREAM_ERROR;
                        break;
                    } else {
                        /* not what the run was parsed with: again, from its start, with this joint
                         * and as many speculated ones as it has confirmed ones by now */
                        run->sched.resize(si);
                        run->sched.push_back(j);
                        run->confirmed = si + 1;
                        run->n = j.new_n;
                        /* what happened once tends to happen again: the same kind of joint, as
                         * many times as the run has confirmed joints */
                        if (j.kind == 0u && run->no_guess_at != si + 1 &&
                            (uint64_t)j.pos + ZD_MIN_LOOKAHEAD <= j.new_n - take)
                            speculate(*run, run->confirmed, 0u); /* only what the parsers can fold */
                        else
                            speculate(*run, run->confirmed, 1u);
                        return run;
                    }
                }
            }
            err = deflate(left_src == 0);
            if (retract)
                return retract;
        }
        if (err == SEC_Z_BUF_ERROR && run && !finishing && produced > run_out0)
            piece(SEC_PIECE_RUN, run_out0, produced - run_out0); /* the caller keeps what fitted */
        s.status = err == SEC_Z_STREAM_END ? SEC_Z_OK : err;
        s.broken_line = broken_line;
        s.delivered = delivered;
        s.produced = produced;
        s.pieces.swap(pieces);
        s.done = true;
        return nullptr;
    }
};

/* May the segmented parser take the run?  It parses with a fixed n per phase, so a joint of
 * kind 0 -- n grows at a cut in the middle of a phase -- must be one the parse up to the cut
 * cannot have noticed: n only matters within MIN_LOOKAHEAD of it (lookahead caps, fill_window
 * calls), so the cut has to lie that far before the old end. */
static inline bool sec_seg_ok(const SecRun &r)
{
    uint32_t n_old = r.n0;
    for (size_t i = 0; i < r.sched.size(); i++) {
        if (r.sched[i].kind == 0u && r.sched[i].pos != ZD_JOINT_ANYWHERE &&
            (uint64_t)r.sched[i].pos + ZD_MIN_LOOKAHEAD > n_old)
            return false;
        n_old = r.sched[i].new_n;
    }
    return true;
}

/* round 0: every section the start of a run; with short sections -- a block or two each, so
 * that every output slice ends in a joint of kind 1 -- parsed together with the next one */
#define SEC_SPECULATE_BELOW 65537u
static inline void sec_first_runs(SecStream &s, uint32_t stream_index, std::vector<SecRun *> &jobs)
{
    if (s.source_len == 0) { /* an empty stream is one empty run */
        SecRun &slot = s.runs[0];
        slot = SecRun();
        slot.stream = stream_index;
        jobs.push_back(&slot);
        return;
    }
    for (uint32_t a = 0; a < s.source_len; a += s.max_block_len) {
        SecRun r;
        r.stream = stream_index;
        r.start = a;
        r.n0 = r.n = std::min(s.max_block_len, s.source_len - a);
        r.more = a + r.n < s.source_len;
        SecRun &slot = s.runs[a];
        slot = r;
        if (s.max_block_len < SEC_SPECULATE_BELOW) {
            SecSim sim(s);
            sim.speculate(slot, 1);
        }
        jobs.push_back(&slot);
    }
}

/* RUNNER: bool operator()(std::vector<SecRun *> &jobs, uint32_t round) parses the runs and fills
 * in their blocks / round / job.  Returns 0, or the runner's error. */
template <class RUNNER>
static inline int sec_compress(std::vector<SecStream> &streams, RUNNER &runner)
{
    std::vector<SecRun *> jobs;
    uint64_t max_rounds = 2;
    for (size_t i = 0; i < streams.size(); i++) {
        sec_first_runs(streams[i], (uint32_t)i, jobs);
        max_rounds = std::max<uint64_t>(max_rounds, 2 * streams[i].runs.size() + 4);
    }
    for (uint32_t round = 0; !jobs.empty(); round++) {
        if (round > max_rounds)
            return SEC_Z_STREAM_ERROR; /* every joint lets a section in: cannot happen */
        const int rc = runner(jobs, round);
        if (rc != 0)
            return rc;
        jobs.clear();
        for (size_t i = 0; i < streams.size(); i++) {
            if (streams[i].done)
                continue;
            SecSim sim(streams[i]);
            SecRun *again = sim.go();
            if (again)
                jobs.push_back(again);
        }
    }
    return 0;
}

#endif
