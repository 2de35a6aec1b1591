function [matches, matchMetric] = matchFeaturesScratch(F1, F2, varargin)
    %MATCHFEATURESSCRATCH Shadows PP/featureMatching/matchFeaturesScratch.m for float descriptors.
    %   Name/value pairs as the reference (:59-76).  'Exhaustive' runs the fused device matcher (2-NN, ratio, threshold,
    %   uniqueness).  'Approximate' with ApproxFloatNNMethod = 'pca2nn' runs nearest2ApproxFloatFast (:442-573) on the
    %   device (PCA-48 of F2, cosine 2-NN) and the filter of :170-211 here; 'kdtree' and 'subsetpdist2' are served by the
    %   exact device search (a kd-tree's knnsearch is exact; the subset draw of :377 is unseeded in the reference).
    %   Binary descriptors go through nearest2HammingExhaustiveMEX.
    p = inputParser; p.KeepUnmatched = true;
    addParameter(p, 'Method', 'Exhaustive');
    addParameter(p, 'MatchThreshold', 3.5);
    addParameter(p, 'MaxRatio', 0.6);
    addParameter(p, 'Unique', true);
    addParameter(p, 'ApproxFloatNNMethod', 'pca2nn');
    parse(p, varargin{:});
    o = p.Results;
    if strcmpi(o.Method, 'Approximate') && strcmpi(o.ApproxFloatNNMethod, 'pca2nn') && ~isempty(F1) && ~isempty(F2)
        A = single(F1); B = single(F2);
        if max(abs(A(:))) > 2 || max(abs(B(:))) > 2        % :105-110, normalizeRowsL2 :232-233
            A = A ./ (sqrt(sum(A .^ 2, 2)) + eps('single'));
            B = B ./ (sqrt(sum(B .^ 2, 2)) + eps('single'));
        end
        [idx2, dBest, dSecond] = aps_mex('pca_2nn', A, B, 48, 1);    % optsLocal of :144-148
        keep = (dBest <= o.MaxRatio * o.MaxRatio * dSecond) & (dBest <= o.MatchThreshold) & isfinite(dBest) & isfinite(dSecond);
        i1 = find(keep); i2 = idx2(keep); d = dBest(keep);
        if o.Unique && ~isempty(i1)                         % rows are distinct already: first come, first served per column
            [d, order] = sort(d, 'ascend');
            i1 = i1(order); i2 = i2(order);
            [~, first] = unique(i2, 'first');
            first = sort(first);
            i1 = i1(first); i2 = i2(first); d = d(first);
        end
        matches = [uint32(i1), uint32(i2)];
        matchMetric = d(:);
        return;
    end
    opts = struct('MaxRatio', o.MaxRatio, 'MatchThreshold', o.MatchThreshold, 'Unique', double(o.Unique));
    [matches, matchMetric] = aps_mex('match_features', single(F1), single(F2), opts);
end
