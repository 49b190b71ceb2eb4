//! Reference-side pinning of the search() boundary (part 2 of tools/ref_golden in the MI355X build's repository).
//!
//! Copy this file to `next-plaid/tests/golden_search.rs` of a lightonai/next-plaid checkout (v1.6.1) and run
//!
//!     NEXT_PLAID_GOLDEN=/path/to/tools/ref_golden/out cargo test --release --test golden_search -- --nocapture
//!
//! `out/` comes from `python tools/ref_golden/export_golden.py`: `index/` is a 2 000-document seeded index in the crate's
//! on-disk format, `golden.json` holds six search cases (dense and batched probe, threshold on / off, subset) with the ids and
//! scores the MI355X build's CPU oracle -- and, bit for bit in ids and to 2e-5 in scores, its HIP path -- returns.  The test runs
//! the same cases through the crate's own `MmapIndex::load` (index.rs:1026) and `MmapIndex::search` (index.rs:1258 ->
//! search.rs:327-640) and compares: identical ids except inside near-ties closer than the tolerance, scores within `rtol`.
//! Passing pins the oracle (and with it every parity claim of the HIP path) to the reference at the search() boundary.

use ndarray::Array2;
use next_plaid::index::MmapIndex;
use next_plaid::SearchParameters;
use serde_json::Value;
use std::path::PathBuf;

fn golden_dir() -> PathBuf {
    PathBuf::from(std::env::var("NEXT_PLAID_GOLDEN").expect("set NEXT_PLAID_GOLDEN to tools/ref_golden/out"))
}

fn query(v: &Value) -> Array2<f32> {
    let rows = v.as_array().unwrap();
    let d = rows[0].as_array().unwrap().len();
    let flat: Vec<f32> = rows
        .iter()
        .flat_map(|r| r.as_array().unwrap().iter().map(|x| x.as_f64().unwrap() as f32))
        .collect();
    Array2::from_shape_vec((rows.len(), d), flat).unwrap()
}

#[test]
fn golden_search_cases_match_the_reference() {
    let dir = golden_dir();
    let doc: Value = serde_json::from_reader(std::fs::File::open(dir.join("golden.json")).unwrap()).unwrap();
    let index = MmapIndex::load(dir.join("index").to_str().unwrap()).expect("MmapIndex::load of the exported index");
    assert_eq!(index.num_documents() as u64, doc["index"]["num_documents"].as_u64().unwrap());
    assert_eq!(index.num_partitions() as u64, doc["index"]["num_partitions"].as_u64().unwrap());
    assert_eq!(index.embedding_dim() as u64, doc["index"]["embedding_dim"].as_u64().unwrap());
    let rtol = doc["rtol"].as_f64().unwrap();
    let queries: Vec<Array2<f32>> = doc["query_tokens"].as_array().unwrap().iter().map(query).collect();
    let mut checked = 0usize;
    for case in doc["cases"].as_array().unwrap() {
        let name = case["name"].as_str().unwrap();
        let p = &case["params"];
        let params = SearchParameters {
            batch_size: p["batch_size"].as_u64().unwrap() as usize,
            n_full_scores: p["n_full_scores"].as_u64().unwrap() as usize,
            top_k: p["top_k"].as_u64().unwrap() as usize,
            n_ivf_probe: p["n_ivf_probe"].as_u64().unwrap() as usize,
            centroid_batch_size: p["centroid_batch_size"].as_u64().unwrap() as usize,
            centroid_score_threshold: p["centroid_score_threshold"].as_f64().map(|x| x as f32),
        };
        let subset: Option<Vec<i64>> = case["subset"]
            .as_array()
            .map(|a| a.iter().map(|x| x.as_i64().unwrap()).collect());
        for (qi, (q, want)) in queries.iter().zip(case["queries"].as_array().unwrap()).enumerate() {
            let got = index.search(q, &params, subset.as_deref()).expect("search");
            let ids: Vec<i64> = want["ids"].as_array().unwrap().iter().map(|x| x.as_i64().unwrap()).collect();
            let scores: Vec<f64> = want["scores"].as_array().unwrap().iter().map(|x| x.as_f64().unwrap()).collect();
            assert_eq!(got.passage_ids.len(), ids.len(), "{name} q{qi}: result count");
            for (r, (&g, &w)) in got.scores.iter().zip(scores.iter()).enumerate() {
                let tol = rtol * w.abs().max(1.0);
                assert!((g as f64 - w).abs() <= tol, "{name} q{qi} rank {r}: score {g} vs {w}");
            }
            for (r, (&g, &w)) in got.passage_ids.iter().zip(ids.iter()).enumerate() {
                if g != w {
                    // only legitimate inside a near-tie: the id must appear in the expected list at a rank whose score is
                    // within the tolerance of this rank's (GEMM summation order may swap such neighbours)
                    let j = ids.iter().position(|&x| x == g);
                    let tol = 2.0 * rtol * scores[r].abs().max(1.0);
                    match j {
                        Some(j) => assert!((scores[j] - scores[r]).abs() <= tol, "{name} q{qi} rank {r}: id {g} vs {w} is not a near-tie"),
                        None => assert!((got.scores[r] as f64 - scores[scores.len() - 1]).abs() <= tol, "{name} q{qi} rank {r}: id {g} not expected"),
                    }
                }
            }
            checked += 1;
        }
    }
    println!("golden_search: {checked} (case, query) pairs match the reference");
    assert_eq!(checked, 36);
}
