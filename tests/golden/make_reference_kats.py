"""Writes tests/golden/reference_kats.json.

The fixture is DATA transcribed from the reference's own unit tests: the
literal inputs each test feeds the hot path and the assertion it makes
(value + tolerance, first id, ordering).  Nothing here is computed by the
oracle or by this repo's kernels, so the file can pin both.  Citations are
file:line in /root/reference (mmailhos/vectorlite v0.1.5).

Run:  python tests/golden/make_reference_kats.py
"""
import json
import os

T = 1e-10  # the tolerance the reference's asserts use

metric_kats = [
    # src/lib.rs:578-650
    dict(src="src/lib.rs:579-583", metric="cosine", a=[1, 2, 3], b=[1, 2, 3], expect=1.0, tol=T),
    dict(src="src/lib.rs:586-590", metric="cosine", a=[1, 0], b=[0, 1], expect=0.0, tol=T),
    dict(src="src/lib.rs:593-597", metric="cosine", a=[1, 2, 3], b=[-1, -2, -3], expect=-1.0, tol=T),
    dict(src="src/lib.rs:600-604", metric="euclidean", a=[1, 2, 3], b=[1, 2, 3], expect=1.0, tol=T),
    dict(src="src/lib.rs:607-612", metric="euclidean", a=[0, 0], b=[3, 4], expect=1.0 / 6.0, tol=T),
    dict(src="src/lib.rs:615-619", metric="manhattan", a=[1, 2, 3], b=[1, 2, 3], expect=1.0, tol=T),
    dict(src="src/lib.rs:622-627", metric="manhattan", a=[0, 0], b=[3, 4], expect=0.125, tol=T),
    dict(src="src/lib.rs:630-635", metric="dotproduct", a=[1, 2, 3], b=[1, 2, 3], expect=14.0, tol=T),
    dict(src="src/lib.rs:638-642", metric="dotproduct", a=[1, 0], b=[0, 1], expect=0.0, tol=T),
    dict(src="src/lib.rs:645-650", metric="dotproduct", a=[1, 2, 3], b=[-1, -2, -3], expect=-14.0, tol=T),
]

basis3 = [[1, 0, 0], [0, 1, 0], [0, 0, 1]]
flat_kats = [
    # each: rows (id, values) in insertion order, query, k, metric, and what the test asserts
    dict(src="src/index/flat.rs:145-184", dim=3, ids=[1, 2, 3], rows=basis3, query=[1.1, 0.1, 0.1], k=2,
         metric="cosine", len=2, first_id=1, first_score_gt=0.99, sorted_desc=True),
    dict(src="src/index/flat.rs:187-201", dim=3, ids=[1, 2, 3], rows=basis3, query=[1, 0, 0], k=2,
         metric="cosine", len=2, first_id=1, first_score=1.0, tol=T),
    dict(src="src/index/flat.rs:204-218", dim=2, ids=[1, 2, 3], rows=[[0, 0], [3, 4], [6, 8]], query=[0, 0], k=2,
         metric="euclidean", len=2, first_id=1, first_score=1.0, tol=T),
    dict(src="src/index/flat.rs:221-235", dim=2, ids=[1, 2, 3], rows=[[0, 0], [3, 4], [6, 8]], query=[0, 0], k=2,
         metric="manhattan", len=2, first_id=1, first_score=1.0, tol=T),
    dict(src="src/index/flat.rs:238-252", dim=2, ids=[1, 2, 3], rows=[[1, 2], [2, 1], [0, 0]], query=[1, 2], k=2,
         metric="dotproduct", len=2, first_id=1, first_score=5.0, tol=T),
    dict(src="src/index/flat.rs:255-274 (cosine)", dim=2, ids=[1, 2], rows=[[1, 2], [2, 1]], query=[1, 2], k=1,
         metric="cosine", len=1, first_id=1),
    dict(src="src/index/flat.rs:255-274 (dot)", dim=2, ids=[1, 2], rows=[[1, 2], [2, 1]], query=[1, 2], k=1,
         metric="dotproduct", len=1, first_id=1),
    dict(src="src/lib.rs:681-693", dim=3, ids=[0, 1, 2], rows=basis3, query=[1, 0, 0], k=2,
         metric="cosine", len=2, first_id=0, first_score=1.0, tol=T),
    dict(src="src/lib.rs:696-723", dim=3, ids=[1, 2], rows=[[1, 0, 0], [0, 1, 0]], query=[1.1, 0.1, 0.1], k=1,
         metric="cosine", len=1, first_id=1),
    dict(src="src/persistence.rs:247-249", dim=3, ids=[0, 1], rows=[[1, 2, 3], [4, 5, 6]], query=[1.1, 2.1, 3.1],
         k=1, metric="cosine", len=1, first_id=0),
    # MockEmbeddingFunction returns vec![1.0; dim] for every text (src/client.rs:504-523): all rows tie,
    # and the test pins id 0 first => stable tie-break by insertion order (src/client.rs:636-667).
    dict(src="src/client.rs:636-667", dim=3, ids=[0, 1], rows=[[1, 1, 1], [1, 1, 1]], query=[1, 1, 1], k=1,
         metric="cosine", len=1, first_id=0),
]

# flat.rs:255-274 also asserts the cosine and dot scores of that index differ.
flat_pairs_differ = [dict(src="src/index/flat.rs:273", a="src/index/flat.rs:255-274 (cosine)",
                          b="src/index/flat.rs:255-274 (dot)")]

conversion_kats = [
    # src/index/hnsw.rs:808-1032  convert_distance_to_similarity(distance, metric)
    dict(src="src/index/hnsw.rs:815-817", metric="euclidean", distance=0.0, expect=1.0, tol=0.0),
    dict(src="src/index/hnsw.rs:820-823", metric="euclidean", distance=0.5, expect=1.0 / 1.5, tol=T),
    dict(src="src/index/hnsw.rs:826-829", metric="euclidean", distance=1.0, expect=0.5, tol=T),
    dict(src="src/index/hnsw.rs:832-835", metric="euclidean", distance=10.0, expect=1.0 / 11.0, tol=T),
    dict(src="src/index/hnsw.rs:838-840", metric="euclidean", distance=100.0, gt=0.0, lt=0.01),
    dict(src="src/index/hnsw.rs:846-848", metric="cosine", distance=0.0, expect=1.0, tol=0.0),
    dict(src="src/index/hnsw.rs:851-854", metric="cosine", distance=100.0, expect=0.9, tol=T),
    dict(src="src/index/hnsw.rs:857-860", metric="cosine", distance=500.0, expect=0.5, tol=T),
    dict(src="src/index/hnsw.rs:863-867", metric="cosine", distance=2000.0, expect=-1.0, tol=T),
    dict(src="src/index/hnsw.rs:875-877", metric="manhattan", distance=0.0, expect=1.0, tol=0.0),
    dict(src="src/index/hnsw.rs:880-883", metric="manhattan", distance=1.0, expect=0.5, tol=T),
    dict(src="src/index/hnsw.rs:886-889", metric="manhattan", distance=5.0, expect=1.0 / 6.0, tol=T),
    dict(src="src/index/hnsw.rs:892-895", metric="manhattan", distance=20.0, expect=1.0 / 21.0, tol=T),
    dict(src="src/index/hnsw.rs:901-903", metric="dotproduct", distance=0.0, expect=1.0, tol=0.0),
    dict(src="src/index/hnsw.rs:906-910", metric="dotproduct", distance=100.0, expect=0.9, tol=T),
    dict(src="src/index/hnsw.rs:913-917", metric="dotproduct", distance=500.0, expect=0.5, tol=T),
    dict(src="src/index/hnsw.rs:920-922", metric="dotproduct", distance=2000.0, expect=0.0, tol=0.0),
    dict(src="src/index/hnsw.rs:1014-1016", metric="cosine", distance=2000.0, expect=-1.0, tol=0.0),
    dict(src="src/index/hnsw.rs:1020-1022", metric="dotproduct", distance=2000.0, expect=0.0, tol=0.0),
    dict(src="src/index/hnsw.rs:1026-1031 (euclid)", metric="euclidean", distance=1000.0, gt=0.0, lt=0.01),
    dict(src="src/index/hnsw.rs:1026-1031 (manhattan)", metric="manhattan", distance=1000.0, gt=0.0, lt=0.01),
]
for m in ("euclidean", "cosine", "manhattan", "dotproduct"):
    conversion_kats.append(dict(src="src/index/hnsw.rs:956-968", metric=m, distance=0.0001, gt=0.9, le=1.0))
for m in ("euclidean", "manhattan"):
    conversion_kats.append(dict(src="src/index/hnsw.rs:969-978", metric=m, distance=100000.0, gt=0.0, lt=0.01))
for d in (0.0, 100.0, 500.0, 1000.0, 1500.0, 2000.0):
    conversion_kats.append(dict(src="src/index/hnsw.rs:925-929", metric="dotproduct", distance=d, ge=0.0, le=1.0))

conversion_monotone = dict(src="src/index/hnsw.rs:933-953", distances=[0.0, 0.5, 1.0, 2.0, 5.0, 10.0],
                           metrics=["euclidean", "cosine", "manhattan"], start=1.0)

# The ingest step (src/embeddings.rs:169-181).  The reference's own check needs its BERT model; what it asserts about
# ANY output of generate_embedding is transcribed here as a property: the in-order f64 norm of the row is 1 +- 1e-10.
embedding_kats = [
    dict(src="src/embeddings.rs:374-383", property="l2_norm_is_one", tol=T),
]

hnsw_search_kats = [
    # What the reference's HNSW tests pin (Euclidean only): first id on well-separated points,
    # result-count bounds and descending order.  Any correct nearest-neighbour walk satisfies them.
    dict(src="src/index/hnsw.rs:562-592", dim=3, metric="euclidean", ids=[1, 2, 3, 4],
         rows=[[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0]], query=[1.1, 0.1, 0.1], k=2,
         nonempty=True, max_len=2, sorted_desc=True),
    dict(src="src/index/hnsw.rs:605-634", dim=3, metric="euclidean", ids=[100, 200, 300, 400],
         rows=[[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0]], query=[1.1, 0.1, 0.1], k=2,
         nonempty=True, max_len=2, first_id=100),
    dict(src="src/index/hnsw.rs:595-601", dim=3, metric="euclidean", ids=[], rows=[], query=[1, 2, 3], k=5,
         empty=True),
]

out = dict(
    _about="Known-answer data transcribed from the reference's unit tests (inputs + asserted outcomes); "
           "see make_reference_kats.py",
    metric_kats=metric_kats, flat_kats=flat_kats, flat_pairs_differ=flat_pairs_differ,
    conversion_kats=conversion_kats, conversion_monotone=conversion_monotone,
    hnsw_search_kats=hnsw_search_kats,
    embedding_kats=embedding_kats,
)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path)
