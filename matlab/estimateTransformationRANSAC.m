function [model, inliers, isFound] = estimateTransformationRANSAC(matchedPoints1, matchedPoints2, transformType, input)
    %ESTIMATETRANSFORMATIONRANSAC Shadows PP/imageMatching/estimateTransformationRANSAC.m ('projective').
    %   The 4-point draws are generated here with randperm exactly as the reference does (:96) and handed to
    %   the device as an explicit input; fitting, scoring, the adaptive stop and the refit run in aps_mex.
    if nargin < 4, input = struct('maxDistance', 2.0, 'inliersConfidence', 99.9, 'maxIter', 500); end
    if ~strcmpi(transformType, 'projective')
        % 'affine' | 'similarity' | 'rigid' | 'translation' (estimateTransformationRANSAC.m:227-439): the reference's own
        % host code runs (inputs.m:74 defaults to 'projective'; only that estimator is built on the device)
        [model, inliers, isFound] = aps_call_shadowed('estimateTransformationRANSAC', mfilename('fullpath'), ...
            matchedPoints1, matchedPoints2, transformType, input);
        return;
    end
    M = size(matchedPoints1, 1);
    if M < 4
        model = []; inliers = false(M, 1); isFound = false; return;
    end
    S = input.maxIter + 64;
    sampleIdx = zeros(4, S, 'uint32');
    for s = 1:S, sampleIdx(:, s) = uint32(randperm(M, 4)); end
    [model, inliers, isFound] = aps_mex('ransac_homography', double(matchedPoints1), double(matchedPoints2), input, sampleIdx);
    if ~isFound, model = []; end
end
