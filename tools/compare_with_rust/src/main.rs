//! compare_with_rust -- the reference's own arithmetic (crate `anofox-forecast` 0.15.3) on this repo's test inputs.
//!
//!   compare_with_rust <cases.json> <fixtures.json>
//!
//! `cases.json` is written by `make_cases.py` (same directory): a list of
//!   { "id", "model", "values": [...], "horizon", "period", "ets_model"?, "model_pool"? }.
//! Every case goes through the crate exactly the way the reference's wrapper drives it
//! (crates/anofox-fcst-core/src/forecast.rs: SES/Holt/HoltWinters/SeasonalES :1102-1144,1206-1232,
//! ETS(spec) :1340-1389, AutoARIMA :1435-1521, AutoETS :1543-1641; hourly fake timestamps :2226-2234),
//! and the point forecasts, the model name as the wrapper would render it, the crate's `Explanation`
//! (serde JSON, when the model is `Inspectable`) and the wall time are written to `fixtures.json`,
//! the format tests/test_rust_fixtures.py consumes.  Comparison style: crates/anofox-fcst-ffi/tests/core_ffi_parity.rs:213-231.
use std::time::Instant;

use anofox_forecast::core::TimeSeries;
use anofox_forecast::models::arima::{AutoARIMA, AutoARIMAConfig};
use anofox_forecast::models::exponential::{
    AutoETS, AutoETSConfig, ETSSpec, HoltLinearTrend, HoltWinters, ModelPool, SeasonalES, SeasonalType,
    SimpleExponentialSmoothing, ETS,
};
use anofox_forecast::models::Inspectable;
use anofox_forecast::prelude::Forecaster;
use chrono::{Duration, TimeZone, Utc};
use serde::{Deserialize, Serialize};

#[derive(Deserialize)]
struct Case {
    id: String,
    model: String,
    values: Vec<f64>,
    horizon: usize,
    #[serde(default)]
    period: usize,
    #[serde(default)]
    ets_model: Option<String>,
    #[serde(default)]
    model_pool: Option<String>,
}

#[derive(Serialize)]
struct Fixture {
    id: String,
    model: String,
    horizon: usize,
    period: usize,
    ets_model: Option<String>,
    model_pool: Option<String>,
    ok: bool,
    error: Option<String>,
    model_name: String,
    point: Vec<f64>,
    explanation: Option<serde_json::Value>,
    seconds: f64,
}

/// forecast.rs:2226-2234
fn make_timeseries(values: &[f64]) -> Result<TimeSeries, String> {
    let base = Utc.with_ymd_and_hms(2024, 1, 1, 0, 0, 0).unwrap();
    let ts: Vec<_> = (0..values.len()).map(|i| base + Duration::hours(i as i64)).collect();
    TimeSeries::univariate(ts, values.to_vec()).map_err(|e| format!("TimeSeries: {e}"))
}

/// forecast.rs:1524-1537
fn parse_pool(s: &str) -> Result<ModelPool, String> {
    match s.to_lowercase().replace(['-', '_'], "").as_str() {
        "complete" => Ok(ModelPool::Complete),
        "nomultiplicativetrend" => Ok(ModelPool::NoMultiplicativeTrend),
        "dampedtrendonly" => Ok(ModelPool::DampedTrendOnly),
        "matcherrorseasonal" => Ok(ModelPool::MatchErrorSeasonal),
        "reduced" => Ok(ModelPool::Reduced),
        _ => Err(format!("Unknown model_pool '{s}'")),
    }
}

fn predict(model: &dyn Forecaster, h: usize) -> Result<Vec<f64>, String> {
    Ok(model.predict(h).map_err(|e| format!("predict: {e}"))?.primary().to_vec())
}

fn explain<M: Inspectable>(m: &M) -> Option<serde_json::Value> {
    Inspectable::explanation(m).ok().and_then(|e| serde_json::to_value(&e).ok())
}

fn run(c: &Case) -> Result<(String, Vec<f64>, Option<serde_json::Value>), String> {
    let ts = make_timeseries(&c.values)?;
    let h = c.horizon;
    let period = if c.period > 0 { c.period } else { 1 };
    match c.model.as_str() {
        "SES" => {
            let mut m = SimpleExponentialSmoothing::new(0.3);
            m.fit(&ts).map_err(|e| e.to_string())?;
            Ok(("SES".into(), predict(&m, h)?, None))
        }
        "SESOptimized" => {
            let mut m = SimpleExponentialSmoothing::auto();
            m.fit(&ts).map_err(|e| e.to_string())?;
            Ok(("SESOptimized".into(), predict(&m, h)?, None))
        }
        "Holt" => {
            let mut m = HoltLinearTrend::auto();
            m.fit(&ts).map_err(|e| e.to_string())?;
            Ok(("Holt".into(), predict(&m, h)?, None))
        }
        "HoltWinters" => {
            let mut m = HoltWinters::auto(period.max(2), SeasonalType::Additive);
            m.fit(&ts).map_err(|e| e.to_string())?;
            Ok(("HoltWinters".into(), predict(&m, h)?, None))
        }
        "SeasonalES" => {
            let mut m = SeasonalES::new(period.max(2));
            m.fit(&ts).map_err(|e| e.to_string())?;
            Ok(("SeasonalES".into(), predict(&m, h)?, None))
        }
        "SeasonalESOptimized" => {
            let mut m = SeasonalES::optimized(period.max(2));
            m.fit(&ts).map_err(|e| e.to_string())?;
            Ok(("SeasonalESOptimized".into(), predict(&m, h)?, None))
        }
        "ETS" => {
            // forecast.rs:1278-1389 (explicit spec only; the spec-less chain is HoltWinters / Holt / SES above)
            let notation = c.ets_model.as_deref().ok_or("ETS case without ets_model")?;
            let spec = ETSSpec::from_notation(notation).map_err(|e| format!("from_notation: {e}"))?;
            if !spec.is_valid() {
                return Err(format!("ETS model '{notation}' is an unstable combination"));
            }
            let sp = if spec.has_seasonal() && period > 1 { period } else { 1 };
            let mut m = ETS::new(spec, sp);
            m.fit(&ts).map_err(|e| format!("Failed to fit ETS model: {e}"))?;
            let point = m.predict(h).map_err(|e| e.to_string())?.point().first().cloned().unwrap_or_default();
            Ok((format!("ETS({})", spec.short_name()), point, explain(&m)))
        }
        "AutoETS" => {
            // forecast.rs:1543-1641
            let mut cfg = if period > 1 { AutoETSConfig::with_period(period) } else { AutoETSConfig::non_seasonal() };
            if let Some(p) = c.model_pool.as_deref() {
                cfg = cfg.with_model_pool(parse_pool(p)?);
            }
            let mut m = AutoETS::with_config(cfg);
            m.fit(&ts).map_err(|e| format!("AutoETS fit failed: {e}"))?;
            let point = predict(&m, h)?;
            let name = match m.selected_spec() {
                Some(s) => format!("AutoETS({:?},{:?},{:?})", s.error, s.trend, s.seasonal),
                None => "AutoETS".to_string(),
            };
            Ok((name, point, explain(&m)))
        }
        "AutoARIMA" => {
            // forecast.rs:1435-1521
            let cfg = if period > 1 { AutoARIMAConfig::default().with_seasonal_period(period) } else { AutoARIMAConfig::default() };
            let mut m = AutoARIMA::with_config(cfg);
            m.fit(&ts).map_err(|e| format!("AutoARIMA fit failed: {e}"))?;
            let point = predict(&m, h)?;
            let name = if let Some(o) = m.selected_full_order() {
                if o.is_seasonal() {
                    format!("AutoARIMA({},{},{})({},{},{})[{}]", o.p, o.d, o.q, o.cap_p, o.cap_d, o.cap_q, o.s)
                } else {
                    format!("AutoARIMA({},{},{})", o.p, o.d, o.q)
                }
            } else if let Some((p, d, q)) = m.selected_order() {
                format!("AutoARIMA({p},{d},{q})")
            } else {
                "AutoARIMA".to_string()
            };
            Ok((name, point, explain(&m)))
        }
        other => Err(format!("model '{other}' is not on the ts_forecast_by hot path")),
    }
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    if args.len() != 3 {
        eprintln!("usage: compare_with_rust <cases.json> <fixtures.json>");
        std::process::exit(2);
    }
    let cases: Vec<Case> = serde_json::from_str(&std::fs::read_to_string(&args[1]).expect("read cases")).expect("parse cases");
    let mut out = Vec::with_capacity(cases.len());
    for c in &cases {
        let t0 = Instant::now();
        // the wrapper guards AutoETS with catch_unwind (forecast.rs:1554, 1631-1640): a panic is a failed case here
        let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| run(c)));
        let seconds = t0.elapsed().as_secs_f64();
        let (ok, error, model_name, point, explanation) = match r {
            Ok(Ok((n, p, e))) => (true, None, n, p, e),
            Ok(Err(e)) => (false, Some(e), String::new(), vec![], None),
            Err(_) => (false, Some("panic".to_string()), String::new(), vec![], None),
        };
        out.push(Fixture {
            id: c.id.clone(), model: c.model.clone(), horizon: c.horizon, period: c.period, ets_model: c.ets_model.clone(),
            model_pool: c.model_pool.clone(), ok, error, model_name, point, explanation, seconds,
        });
    }
    let doc = serde_json::json!({
        "crate": "anofox-forecast", "version": "0.15.3",
        "note": "generated by tools/compare_with_rust; inputs regenerate from make_cases.py (ids are stable)",
        "fixtures": out,
    });
    std::fs::write(&args[2], serde_json::to_string(&doc).unwrap()).expect("write fixtures");
    let total: f64 = doc["fixtures"].as_array().unwrap().iter().map(|f| f["seconds"].as_f64().unwrap()).sum();
    eprintln!("{} cases, {:.2} s in the crate", cases.len(), total);
}
